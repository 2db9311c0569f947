// Cost of a dependent kernel boundary: N small kernels back to back on one stream, launched one by one
// and as one captured hipGraph.  (scratch; not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
__global__ void small(double *p, int j) { if (threadIdx.x == 0) p[blockIdx.x] += j; }
int main()
{
    double *d; CK(hipMalloc(&d, 8 * 4096)); CK(hipMemset(d, 0, 8 * 4096));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int const N = 3000;
    for (int grid : {1, 313, 1280}) {
        for (int threads : {256, 1024}) {
            for (int w = 0; w < 2; w++) {
                CK(hipEventRecord(e0, s));
                for (int j = 0; j < N; j++) hipLaunchKernelGGL(small, dim3(grid), dim3(threads), 0, s, d, j);
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            hipGraph_t g; hipGraphExec_t ge;
            auto t0 = std::chrono::steady_clock::now();
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int j = 0; j < N; j++) hipLaunchKernelGGL(small, dim3(grid), dim3(threads), 0, s, d, j);
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            double cap = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            float msg = 0;
            for (int w = 0; w < 2; w++) {
                CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&msg, e0, e1));
            }
            printf("grid %4d x %4d threads: stream launches %.2f us per kernel; graph %.2f us per kernel (capture + instantiate %.1f ms for %d nodes)\n",
                grid, threads, ms * 1e3 / N, msg * 1e3 / N, cap * 1e3, N);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
