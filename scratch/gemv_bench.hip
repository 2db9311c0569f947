// Microbenchmark of panel-gemv variants (scratch; not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

template <int UNROLL, bool NT, int RPL /*row pairs per lane*/>
__global__ __launch_bounds__(256)
void gemv_k(double const *__restrict__ A, int ldA, double const *__restrict__ v,
    int m, int ncols, int cps, double *__restrict__ ypart, int ldy)
{
    constexpr int ROWS = 512 * RPL;
    int const g = blockIdx.x * ROWS + threadIdx.x * 2;
    int const c_begin = blockIdx.y * cps;
    int const c_end = min(ncols, c_begin + cps);
    if (g >= m) return;
    double acc[RPL][2][2] = {};
    double const *a = A + (size_t)c_begin * ldA + g;
    int c = c_begin;
    for (; c + UNROLL <= c_end; c += UNROLL) {
        d2 x[UNROLL][RPL];
        #pragma unroll
        for (int u = 0; u < UNROLL; u++)
            #pragma unroll
            for (int q = 0; q < RPL; q++) {
                d2 const *p = reinterpret_cast<d2 const *>(a + (size_t)u * ldA + q * 512);
                x[u][q] = NT ? __builtin_nontemporal_load(p) : *p;
            }
        #pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            double vv = v[c + u];
            #pragma unroll
            for (int q = 0; q < RPL; q++) {
                acc[q][u & 1][0] += x[u][q].x * vv;
                acc[q][u & 1][1] += x[u][q].y * vv;
            }
        }
        a += (size_t)UNROLL * ldA;
    }
    for (int q = 0; q < RPL; q++) {
        int gg = g + q * 512;
        if (gg < m) {
            ypart[(size_t)blockIdx.y * ldy + gg] = acc[q][0][0] + acc[q][1][0];
            ypart[(size_t)blockIdx.y * ldy + gg + 1] = acc[q][0][1] + acc[q][1][1];
        }
    }
}

template <int UNROLL, bool NT, int RPL>
void run(const char *name, double *A, int ld, double *v, double *yp, int m, int ncols, int nsplit)
{
    int cps = ((ncols + nsplit - 1) / nsplit + UNROLL - 1) / UNROLL * UNROLL;
    int ns = (ncols + cps - 1) / cps;
    dim3 grid((m + 512 * RPL - 1) / (512 * RPL), ns);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((gemv_k<UNROLL, NT, RPL>), grid, dim3(256), 0, 0, A, ld, v, m, ncols, cps, yp, ld);
    int reps = 10;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((gemv_k<UNROLL, NT, RPL>), grid, dim3(256), 0, 0, A, ld, v, m, ncols, cps, yp, ld);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double gbs = 8.0 * m * (double)ncols * reps / (ms * 1e-3) / 1e9;
    printf("%-28s m=%6d ncols=%6d nsplit=%3d blocks=%5d  %8.1f us  %7.1f GB/s\n", name, m, ncols, ns, grid.x * grid.y, ms / reps * 1e3, gbs);
}

int main()
{
    int n = 20000, ld = 20000;
    double *A, *v, *yp;
    CK(hipMalloc(&A, (size_t)ld * n * 8)); CK(hipMalloc(&v, n * 8)); CK(hipMalloc(&yp, (size_t)64 * ld * 8));
    CK(hipMemset(A, 0x3c, (size_t)ld * n * 8)); CK(hipMemset(v, 0x3c, n * 8));
    for (int m : {20000, 10000, 4000}) {
        for (int ns : {8, 16, 32, 64}) {
            run<8, false, 1>("u8 rpl1", A, ld, v, yp, m, m, ns);
        }
        run<16, false, 1>("u16 rpl1", A, ld, v, yp, m, m, 32);
        run<4, false, 1>("u4 rpl1", A, ld, v, yp, m, m, 32);
        run<8, true, 1>("u8 nt rpl1", A, ld, v, yp, m, m, 32);
        run<16, true, 1>("u16 nt rpl1", A, ld, v, yp, m, m, 32);
        run<8, false, 2>("u8 rpl2", A, ld, v, yp, m, m, 32);
        run<8, true, 2>("u8 nt rpl2", A, ld, v, yp, m, m, 32);
        run<4, false, 2>("u4 rpl2", A, ld, v, yp, m, m, 32);
        run<8, false, 2>("u8 rpl2 s64", A, ld, v, yp, m, m, 64);
        run<16, true, 1>("u16 nt rpl1 s64", A, ld, v, yp, m, m, 64);
    }
    return 0;
}
