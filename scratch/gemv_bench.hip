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

// persistent variant: blocks pull (row tile, column chunk) work items from an atomic counter;
// partial sums go to ypart[chunk] (one slice per column chunk)
template <int UNROLL, int CHUNK>
__global__ __launch_bounds__(256)
void gemv_dyn(double const *__restrict__ A, int ldA, double const *__restrict__ v,
    int m, int ncols, int row_tiles, int nchunks, double *__restrict__ ypart, int ldy, unsigned *counter)
{
    __shared__ unsigned s_item;
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(counter, 1u);
        __syncthreads();
        unsigned const item = s_item;
        __syncthreads();
        if (item >= (unsigned)(row_tiles * nchunks)) return;
        int const tile = item % row_tiles, chunk = item / row_tiles;
        int const g = tile * 512 + threadIdx.x * 2;
        int const c_begin = chunk * CHUNK, c_end = min(ncols, c_begin + CHUNK);
        if (g >= m) continue;
        double a0 = 0, a1 = 0, b0 = 0, b1 = 0;
        double const *a = A + (size_t)c_begin * ldA + g;
        for (int c = c_begin; c + UNROLL <= c_end; c += UNROLL) {
            d2 x[UNROLL];
            #pragma unroll
            for (int u = 0; u < UNROLL; u++)
                x[u] = __builtin_nontemporal_load(reinterpret_cast<d2 const *>(a + (size_t)u * ldA));
            #pragma unroll
            for (int u = 0; u < UNROLL; u += 2) {
                double v0 = v[c + u], v1 = v[c + u + 1];
                a0 += x[u].x * v0; a1 += x[u].y * v0; b0 += x[u + 1].x * v1; b1 += x[u + 1].y * v1;
            }
            a += (size_t)UNROLL * ldA;
        }
        ypart[(size_t)chunk * ldy + g] = a0 + b0;
        ypart[(size_t)chunk * ldy + g + 1] = a1 + b1;
    }
}
template <int UNROLL, int CHUNK>
void run_dyn(const char *name, double *A, int ld, double *v, double *yp, int m, int ncols, int blocks, unsigned *counter)
{
    int row_tiles = (m + 511) / 512, nchunks = (ncols + CHUNK - 1) / CHUNK;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int reps = 10;
    CK(hipMemset(counter, 0, 4 * 64));
    for (int w = 0; w < 2; w++) { hipLaunchKernelGGL((gemv_dyn<UNROLL, CHUNK>), dim3(blocks), dim3(256), 0, 0, A, ld, v, m, ncols, row_tiles, nchunks, yp, ld, counter + w); }
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((gemv_dyn<UNROLL, CHUNK>), dim3(blocks), dim3(256), 0, 0, A, ld, v, m, ncols, row_tiles, nchunks, yp, ld, counter + 2 + r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double gbs = 8.0 * m * (double)ncols * reps / (ms * 1e-3) / 1e9;
    printf("%-28s m=%6d ncols=%6d chunks=%3d blocks=%5d  %8.1f us  %7.1f GB/s\n", name, m, ncols, nchunks, blocks, ms / reps * 1e3, gbs);
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

// A dependent latency-bound kernel between the gemv launches (what colA/colC are in the
// library): does the gemv run slower right after a quiet phase?
__global__ void gap_kernel(long cycles, double *sink)
{
    long t0 = clock64();
    while (clock64() - t0 < cycles) { __builtin_amdgcn_s_sleep(8); }
    if (sink && threadIdx.x == 9999) *sink = 1.0;
}
template <int UNROLL>
void run_gapped(const char *name, double *A, int ld, double *v, double *yp, int m, int ncols, int nsplit, int gap_blocks, long gap_us)
{
    int cps = ((ncols + nsplit - 1) / nsplit + UNROLL - 1) / UNROLL * UNROLL;
    int ns = (ncols + cps - 1) / cps;
    dim3 grid((m + 511) / 512, ns);
    int reps = 20;
    std::vector<hipEvent_t> ev(2 * reps);
    for (auto &evt : ev) CK(hipEventCreate(&evt));
    for (int r = 0; r < reps; r++) {
        if (gap_blocks > 0) hipLaunchKernelGGL(gap_kernel, dim3(gap_blocks), dim3(256), 0, 0, gap_us * 100, (double *)nullptr);   // 100 MHz clock64
        CK(hipEventRecord(ev[2 * r]));
        hipLaunchKernelGGL((gemv_k<UNROLL, true, 1>), grid, dim3(256), 0, 0, A, ld, v, m, ncols, cps, yp, ld);
        CK(hipEventRecord(ev[2 * r + 1]));
    }
    CK(hipDeviceSynchronize());
    double tot = 0;
    for (int r = 4; r < reps; r++) { float ms; CK(hipEventElapsedTime(&ms, ev[2 * r], ev[2 * r + 1])); tot += ms; }
    double us = tot / (reps - 4) * 1e3;
    printf("%-28s m=%6d gap %3ld us x %4d blocks: gemv %8.1f us  %7.1f GB/s\n", name, m, gap_us, gap_blocks, us, 8.0 * m * (double)ncols / us / 1e3);
}

int main()
{
    int n = 20000, ld = 20000;
    double *A, *v, *yp; unsigned *counter;
    CK(hipMalloc(&A, (size_t)ld * n * 8)); CK(hipMalloc(&v, n * 8)); CK(hipMalloc(&yp, (size_t)320 * ld * 8));
    CK(hipMalloc(&counter, 4 * 64));
    CK(hipMemset(A, 0x3c, (size_t)ld * n * 8)); CK(hipMemset(v, 0x3c, n * 8));
    for (int m : {19000, 11000, 7000}) {
        run_gapped<16>("gapped", A, ld, v, yp, m, m, 32, 0, 0);
        run_gapped<16>("gapped", A, ld, v, yp, m, m, 32, 1, 10);
        run_gapped<16>("gapped", A, ld, v, yp, m, m, 32, 1, 40);
        run_gapped<16>("gapped", A, ld, v, yp, m, m, 32, 300, 40);
        run_gapped<16>("gapped", A, ld, v, yp, m, m, 32, 1, 200);
    }
    for (int m : {3000}) {
        int row_tiles = (m + 511) / 512;
        int want = std::max(1, 1280 / row_tiles);
        int ns = std::min(want, 32);
        run<16, true, 1>("u16 nt (library rule)", A, ld, v, yp, m, m, ns);
        run<16, true, 1>("u16 nt s64", A, ld, v, yp, m, m, std::min(std::max(1, 2560 / row_tiles), 64));
        run<16, true, 1>("u16 nt s128", A, ld, v, yp, m, m, std::min(std::max(1, 5120 / row_tiles), 128));
        run<8, true, 1>("u8 nt", A, ld, v, yp, m, m, ns);
        run<16, false, 1>("u16 temporal", A, ld, v, yp, m, m, ns);
        run_dyn<16, 64>("dyn u16 chunk64 x1024", A, ld, v, yp, m, m, 1024, counter);
        run_dyn<16, 64>("dyn u16 chunk64 x1536", A, ld, v, yp, m, m, 1536, counter);
        run_dyn<16, 128>("dyn u16 chunk128 x1280", A, ld, v, yp, m, m, 1280, counter);
        run_dyn<16, 128>("dyn u16 chunk128 x2048", A, ld, v, yp, m, m, 2048, counter);
    }
    return 0;
}
