// Scratch: what does a cross-stream dependency cost when the waiter really has to wait?  Two streams play ping-pong
// with a short kernel each (stream A: kernel, record; stream B: wait, kernel, record; A: wait ...), for several
// kinds of event and stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin_kernel(long long cycles, double *out)
{
    long long const t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) { }
    if (out) out[0] = 1.0;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int run(const char *name, unsigned evflags, int prio_a, int prio_b, int iters, long long cycles)
{
    hipStream_t a, b;
    CK(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, prio_a));
    CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, prio_b));
    hipEvent_t ea, eb;
    CK(hipEventCreateWithFlags(&ea, evflags)); CK(hipEventCreateWithFlags(&eb, evflags));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipDeviceSynchronize());
        double const t0 = now();
        for (int i = 0; i < iters; i++) {
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, cycles, (double *)nullptr);
            CK(hipEventRecord(ea, a));
            CK(hipStreamWaitEvent(b, ea, 0));
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, cycles, (double *)nullptr);
            CK(hipEventRecord(eb, b));
            CK(hipStreamWaitEvent(a, eb, 0));
        }
        CK(hipDeviceSynchronize());
        double const dt = now() - t0;
        if (rep == 1) printf("%-52s %7.1f us per hand-over (two kernels of %.1f us in it)\n", name, dt / iters / 2 * 1e6, cycles / 100.0 / 1.0);
    }
    // the same kernels on one stream
    CK(hipDeviceSynchronize());
    double const t0 = now();
    for (int i = 0; i < 2 * iters; i++) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, cycles, (double *)nullptr);
    CK(hipDeviceSynchronize());
    printf("%-52s %7.1f us per kernel\n", "  (one stream, no events)", (now() - t0) / iters / 2 * 1e6);
    return 0;
}
int main()
{
    int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    long long const cycles = 1000;      // s_memtime ticks at 100 MHz: 10 us
    run("events: disable timing", hipEventDisableTiming, 0, 0, 2000, cycles);
    run("events: default (timing)", hipEventDefault, 0, 0, 2000, cycles);
    run("events: disable timing + release to device", hipEventDisableTiming | hipEventReleaseToDevice, 0, 0, 2000, cycles);
    run("events: disable timing + release to system", hipEventDisableTiming | hipEventReleaseToSystem, 0, 0, 2000, cycles);
    run("events: disable timing; streams high / normal", hipEventDisableTiming, hi, 0, 2000, cycles);
    run("events: disable timing; streams high / low", hipEventDisableTiming, hi, lo, 2000, cycles);
    return 0;
}
