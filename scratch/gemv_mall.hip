// Does the Infinity Cache keep the tail of a streamed matrix?  The panel gemv reads the same trailing
// matrix once per column; if the last ~256 MB of one pass are still cached, a pass that walks its
// columns in the opposite direction starts on cached data.  (scratch; not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

template <int UNROLL, int MODE /*0 nt, 1 plain, 2 nt on the body + plain on the last TAILC columns walked*/>
__global__ __launch_bounds__(256)
void gemv_k(double const *__restrict__ A, int ldA, double const *__restrict__ v,
    int m, int ncols, int cps, double *__restrict__ ypart, int ldy, int rev, int tailc)
{
    int const g = blockIdx.x * 512 + threadIdx.x * 2;
    int const c_begin = blockIdx.y * cps;
    int const c_end = min(ncols, c_begin + cps);
    if (g >= m) return;
    double a0 = 0, a1 = 0, b0 = 0, b1 = 0;
    int const steps = (c_end - c_begin) / UNROLL;
    for (int s = 0; s < steps; s++) {
        int const c = rev ? c_begin + (steps - 1 - s) * UNROLL : c_begin + s * UNROLL;
        double const *a = A + (size_t)c * ldA + g;
        bool const plain = MODE == 1 || (MODE == 2 && (steps - s) * UNROLL <= tailc);
        d2 x[UNROLL];
        if (plain) {
            #pragma unroll
            for (int u = 0; u < UNROLL; u++) x[u] = *reinterpret_cast<d2 const *>(a + (size_t)u * ldA);
        } else {
            #pragma unroll
            for (int u = 0; u < UNROLL; u++) x[u] = __builtin_nontemporal_load(reinterpret_cast<d2 const *>(a + (size_t)u * ldA));
        }
        #pragma unroll
        for (int u = 0; u < UNROLL; u += 2) {
            double v0 = v[c + u], v1 = v[c + u + 1];
            a0 += x[u].x * v0; a1 += x[u].y * v0; b0 += x[u + 1].x * v1; b1 += x[u + 1].y * v1;
        }
    }
    ypart[(size_t)blockIdx.y * ldy + g] = a0 + b0;
    ypart[(size_t)blockIdx.y * ldy + g + 1] = a1 + b1;
}

template <int MODE>
void run(const char *name, double *A, int ld, double *v, double *yp, int m, int nsplit, bool alternate, int tailc)
{
    constexpr int UNROLL = 16;
    int ncols = m;
    int cps = ((ncols + nsplit - 1) / nsplit + UNROLL - 1) / UNROLL * UNROLL;
    int ns = (ncols + cps - 1) / cps;
    dim3 grid((m + 511) / 512, ns);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((gemv_k<UNROLL, MODE>), grid, dim3(256), 0, 0, A, ld, v, m, ncols, cps, yp, ld, alternate ? (w & 1) : 0, tailc);
    int reps = 10;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((gemv_k<UNROLL, MODE>), grid, dim3(256), 0, 0, A, ld, v, m, ncols, cps, yp, ld, alternate ? (r & 1) : 0, tailc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double gbs = 8.0 * m * (double)ncols * reps / (ms * 1e-3) / 1e9;
    printf("%-34s m=%6d (%5.0f MB) nsplit=%3d  %8.1f us  %7.1f GB/s\n", name, m, 8.0 * m * m / 1e6, ns, ms / reps * 1e3, gbs);
}

int main()
{
    int n = 20000, ld = 20000;
    double *A, *v, *yp;
    CK(hipMalloc(&A, (size_t)ld * n * 8)); CK(hipMalloc(&v, n * 8)); CK(hipMalloc(&yp, (size_t)320 * ld * 8));
    CK(hipMemset(A, 0x3c, (size_t)ld * n * 8)); CK(hipMemset(v, 0x3c, n * 8));
    for (int m : {19000, 15000, 11000, 8000, 6500}) {
        int row_tiles = (m + 511) / 512;
        int ns = std::min(std::max(1, 1280 / row_tiles), 32);
        run<0>("nt, same direction", A, ld, v, yp, m, ns, false, 0);
        run<0>("nt, alternating", A, ld, v, yp, m, ns, true, 0);
        run<1>("plain, same direction", A, ld, v, yp, m, ns, false, 0);
        run<1>("plain, alternating", A, ld, v, yp, m, ns, true, 0);
        for (int mb : {128, 192, 256}) {
            int tailc = (int)((double)mb * 1e6 / (8.0 * m) / ns);     // columns per chunk whose total is mb
            char nm[64]; snprintf(nm, sizeof nm, "nt + plain tail %d MB, alternating", mb);
            run<2>(nm, A, ld, v, yp, m, ns, true, tailc);
        }
    }
    return 0;
}
