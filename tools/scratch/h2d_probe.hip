// Build: hipcc --offload-arch=gfx950 -O2 -o tools/scratch/h2d_probe tools/scratch/h2d_probe.hip ; run: ./tools/scratch/h2d_probe MODE
// H2D copies from a ring of pinned chunks beside a kernel that fills the chip: which variations make the runtime fall back
// from the DMA engines to its copy kernel (18 GB/s beside compute instead of 52)?
// modes: 0 pure H2D stream, 1 a tiny kernel per 16 copies on the copy stream, 2 a memset per 16 copies, 3 waits on copy events,
//        4 pure D2H stream, 5 D2H that waits for a kernel of another stream, 6 H2D that waits for a kernel of another stream
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <time.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void spin(double *o, int n) { double a = threadIdx.x; for (int i = 0; i < n; ++i) a = a * 1.0000001 + 0.5; if (a == 1.5) o[0] = a; }
__global__ void tiny(int *p) { if (threadIdx.x == 999) p[0] = 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    int mode = argc > 1 ? atoi(argv[1]) : 0;
    if (mode == 9) { CK(hipSetDeviceFlags(hipDeviceScheduleBlockingSync)); mode = 7; printf("(hipDeviceScheduleBlockingSync) "); }
    if (mode == 10) { mode = 7; printf("(offset source, event-sync first) "); }
    const size_t CH = 64u << 20; const int NCH = 4, N = 64;
    void *pin[NCH]; hipEvent_t ev[NCH]; char *dev; double *o; int *ip;
    for (int k = 0; k < NCH; ++k) { CK(hipHostMalloc(&pin[k], CH, hipHostMallocDefault)); memset(pin[k], k, CH); CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming)); }
    CK(hipMalloc(&dev, (size_t)N * CH)); CK(hipMalloc(&o, 64)); CK(hipMalloc(&ip, 64));
    hipStream_t cs, ks, ts; hipEvent_t evk;
    CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&ks, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&ts, hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&evk, hipEventDisableTiming));
    auto run = [&](bool busy) {
        CK(hipDeviceSynchronize());
        if (busy) hipLaunchKernelGGL(spin, dim3(256 * 16), dim3(256), 0, ks, o, 3000000);
        const double t = now();
        for (int i = 0; i < N; ++i) {
            const int k = i % NCH;
            if (i >= NCH) CK(hipEventSynchronize(ev[k]));
            if (mode == 1 && i % 16 == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, cs, ip);
            if (mode == 2 && i % 16 == 0) CK(hipMemsetAsync(ip, 0, 64, cs));
            if (mode == 3) CK(hipStreamWaitEvent(cs, ev[(k + 1) % NCH], 0));
            if ((mode == 5 || mode == 6) && i % 16 == 0) { /* the copy waits for a (tiny) kernel of ANOTHER stream */
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, ts, ip);
                CK(hipEventRecord(evk, ts));
                CK(hipStreamWaitEvent(cs, evk, 0));
            }
            if (mode == 4 || mode == 5) CK(hipMemcpyAsync(pin[k], dev + (size_t)i * CH, CH, hipMemcpyDeviceToHost, cs));
            else
            CK(hipMemcpyAsync(dev + (size_t)i * CH, pin[k], CH, hipMemcpyHostToDevice, cs));
            CK(hipEventRecord(ev[k], cs));
        }
        CK(hipStreamSynchronize(cs));
        const double dt = now() - t;
        CK(hipDeviceSynchronize());
        return (double)N * CH / dt / 1e9;
    };
    if (mode == 7 || mode == 8) { /* one 19 MB copy now and then (the packed results of a step), timed one by one */
        const size_t n = 19u << 20;
        for (int busy = 0; busy < 2; ++busy) {
            CK(hipDeviceSynchronize());
            double worst = 0, sum = 0;
            for (int i = 0; i < 12; ++i) {
                if (busy) hipLaunchKernelGGL(spin, dim3(256 * 16), dim3(256), 0, ks, o, 600000);
                struct timespec ts = {0, 20000000}; nanosleep(&ts, nullptr);
                const double t = now();
                if (mode == 7) CK(hipMemcpyAsync(pin[0], dev, n, hipMemcpyDeviceToHost, cs));
                else CK(hipMemcpyAsync(dev, pin[0], n, hipMemcpyHostToDevice, cs));
                CK(hipStreamSynchronize(cs));
                const double dt = now() - t; sum += dt; if (dt > worst) worst = dt;
                CK(hipDeviceSynchronize());
            }
            printf("mode %d, %s: 19 MB copy avg %.2f ms, worst %.2f ms\n", mode, busy ? "beside a chip-filling kernel" : "alone", sum / 12 * 1e3, worst * 1e3);
        }
        return 0;
    }
    run(false);
    const double a = run(false), b = run(true);
    printf("mode %d (GPU_MAX_HW_QUEUES=%s): alone %.1f GB/s, beside a chip-filling kernel %.1f GB/s\n", mode, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "-", a, b);
    return 0;
}
