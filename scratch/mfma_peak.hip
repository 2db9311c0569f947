#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (d4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int i = 0; i < NACC; i++)
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int wgs)
{
    double *out; hipMalloc(&out, (size_t)blocks * 256 * 8);
    int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0, 2.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 2.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * NACC * 2048.0;
    printf("NACC=%2d blocks=%5d (x%d per CU): %.2f ms  %.1f TFLOP/s\n", NACC, blocks, wgs, ms, flops / ms / 1e9);
    hipFree(out);
}
int main()
{
    run<16>(256, 1); run<16>(512, 2); run<16>(1024, 4);
    run<4>(256, 1); run<4>(512, 2); run<8>(256, 1); run<1>(1024, 4); run<2>(2048, 8);
    return 0;
}
