// Latency probes for the chain of the Hessenberg-triangular reduction (one wave, dependent fp64 work).
//   hipcc --offload-arch=gfx950 -O3 -o ht_micro scratch/ht_micro.hip && ./ht_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ inline double readlane_d(double v, int l)
{
    int const lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int const hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__global__ void k_fma(double *out, long long *clk, int iters, double a, double b)
{
    double x = out[threadIdx.x];
    long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) x = fma(x, a, b);
    }
    long long t1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

__global__ void k_rsq(double *out, long long *clk, int iters, double a)
{
    double x = out[threadIdx.x];
    long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) x = __builtin_amdgcn_rsq(x) + a;
    }
    long long t1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

// the rotation of one chain step, fed back through a lane read (the dependent path of ht_chain_kernel)
__global__ void k_step(double *out, long long *clk, int iters)
{
    int const lane = threadIdx.x;
    double y = out[lane], x = 0.3 + 0.01 * lane, fv = 0.2 + 0.003 * lane;
    long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int q = 63; q >= 32; q--) {
            double const d = readlane_d(y, q), f = readlane_d(fv, q);
            double const h2 = fma(d, d, f * f);
            double r = __builtin_amdgcn_rsq(h2);
            double e1 = fma(-h2 * r, r, 1.0);
            r = fma(r * e1, fma(0.375, e1, 0.5), r);
            e1 = fma(-h2 * r, r, 1.0);
            r = fma(r * 0.5, e1, r);
            bool const trivial = f == 0.0;
            double const c = trivial ? 1.0 : fabs(d) * r;
            double const s = trivial ? 0.0 : f * copysign(r, d);
            double const yc = c * x - s * y;
            y = (lane < q) ? yc : y + 1.0;
        }
    }
    long long t1 = clock64(), w1 = wall_clock64();
    out[lane] = y;
    if (lane == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

// the LDS-resident chain loop of ht_chain_kernel in isolation; MODE bits: 1 = LDS reads of the column,
// 2 = LDS writes of the results, 4 = rotation table write, 8 = counter publication every 16, 16 = safe-range branch,
// 32 = three spinning waves next to it
template <int MODE>
__global__ void k_chain(double *out, long long *clk, int iters)
{
    __shared__ double tl[65][64];
    __shared__ double rot[64][2];
    __shared__ volatile int ctr[4];
    __shared__ double sink[128];
    int const lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c = 0; c < 65; c++) tl[c][lane] = 0.3 + 0.01 * lane + 0.001 * c;
    if (threadIdx.x == 0) { ctr[0] = 0; ctr[1] = 0; }
    __syncthreads();
    if (wv > 0) {
        if (MODE & 32) { while (ctr[1] == 0) __builtin_amdgcn_s_sleep(1); }
        return;
    }
    double y = out[lane];
    long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        double xn = tl[63][lane], fn = tl[63][63];
#pragma unroll 1
        for (int q = 63; q >= 0; q--) {
            double x = 0.3 + 0.01 * lane, f = 0.2;
            if (MODE & 1) { x = xn; f = fn; int const qn = q > 0 ? q - 1 : 0; xn = tl[qn][lane]; fn = tl[qn][qn]; }
            double const d = readlane_d(y, q);
            double const h2 = fma(d, d, f * f);
            double c, s, r;
            if ((MODE & 16) && (!(h2 < 1e300) || (h2 < 1e-280 && h2 != 0.0))) {
                int const e = ilogb(fmax(fabs(f), fabs(d)));
                double const f1 = scalbn(d, -e), g1 = scalbn(f, -e);
                double const dd = sqrt(f1 * f1 + g1 * g1);
                c = fabs(f1) / dd; s = g1 / copysign(dd, f1); r = scalbn(copysign(dd, f1), e);
            } else {
                double rr = __builtin_amdgcn_rsq(h2);
                double e1 = fma(-h2 * rr, rr, 1.0);
                rr = fma(rr * e1, fma(0.375, e1, 0.5), rr);
                e1 = fma(-h2 * rr, rr, 1.0);
                rr = fma(rr * 0.5, e1, rr);
                bool const trivial = f == 0.0;
                c = trivial ? 1.0 : fabs(d) * rr;
                s = trivial ? 0.0 : f * copysign(rr, d);
                r = trivial ? d : copysign(h2 * rr, d);
            }
            double const yf = s * x + c * y, yc = c * x - s * y;
            if (MODE & 64) {            // branch-free stores: idle lanes aim at a scratch row
                if (MODE & 4) *(lane == 0 ? (double2 *)rot[q] : (double2 *)&sink[2 * lane]) = double2{c, s};
                if (MODE & 2) {
                    *(lane <= q ? &tl[q + 1][lane] : &sink[lane]) = (lane < q) ? yf : r;
                    *(lane == q ? &tl[q][lane] : &sink[lane]) = 0.1;
                }
            } else {
                if ((MODE & 4) && lane == 0) { rot[q][0] = c; rot[q][1] = s; }
                if (MODE & 2) {
                    if (lane <= q) tl[q + 1][lane] = (lane < q) ? yf : r;
                    if (lane == q) tl[q][lane] = 0.1;
                }
            }
            y = (lane < q) ? yc : (lane == q ? 0.5 : y);
            if ((MODE & 8) && (q % 16 == 0)) {
                asm volatile("" ::: "memory");
                if (lane == 0) ctr[0] = 64 - q;
            }
        }
    }
    long long t1 = clock64(), w1 = wall_clock64();
    out[lane] = y;
    if (lane == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; ctr[1] = 1; }
}

// accuracy of v_rsq_f64 and of one / two Newton steps: max |1 - x y^2| / 2 over a sweep of x
__global__ void k_rsq_err(double *out)
{
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < 20000; i++) {
        double const x = (1.0 + (threadIdx.x * 20000 + i) * (3.0 / (256 * 20000.0))) * ((i & 1) ? 1e-7 : 1e5);
        double y = __builtin_amdgcn_rsq(x);
        double r = fma(-x * y, y, 1.0);
        e0 = fmax(e0, fabs(r) * 0.5);
        y = fma(y * r, fma(0.375, r, 0.5), y);
        r = fma(-x * y, y, 1.0);
        e1 = fmax(e1, fabs(r) * 0.5);
        y = fma(y * 0.5, r, y);
        r = fma(-x * y, y, 1.0);
        e2 = fmax(e2, fabs(r) * 0.5);
    }
    out[3 * threadIdx.x] = e0; out[3 * threadIdx.x + 1] = e1; out[3 * threadIdx.x + 2] = e2;
}

// the production loop: explicit ds instructions, explicit waits (see ht_chain_kernel)
typedef double v2d __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k_chain_asm(double *out, long long *clk, int iters)
{
    __shared__ double tl[65][64];
    __shared__ double rot[64][2];
    __shared__ int ctr[4];
    __shared__ double sink[192];
    int const lane = threadIdx.x & 63;
    for (int c = 0; c < 65; c++) tl[c][lane] = 0.3 + 0.01 * lane + 0.001 * c;
    __syncthreads();
    unsigned const a_tile = (unsigned)(size_t)&tl[0][0], a_rot = (unsigned)(size_t)&rot[0][0], a_sink = (unsigned)(size_t)&sink[0], a_ctr = (unsigned)(size_t)&ctr[0];
    double y = out[lane];
    long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        double fv, xn;
        asm volatile("ds_read_b64 %0, %1" : "=v"(fv) : "v"(a_tile + 8u * 65u * lane) : "memory");
        asm volatile("ds_read_b64 %0, %1" : "=v"(xn) : "v"(a_tile + 8u * (64u * 63 + lane)) : "memory");
        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fv) :: "memory");
#pragma unroll 1
        for (int q = 63; q >= 0; q--) {
            double x = xn;
            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(x) :: "memory");
            asm volatile("ds_read_b64 %0, %1" : "=v"(xn) : "v"(a_tile + 8u * (64u * (q > 0 ? q - 1 : 0) + lane)) : "memory");
            double const d = readlane_d(y, q), f = (MODE & 1) ? readlane_d(fv, q) : 0.2;
            double const h2 = fma(d, d, f * f);
            double rr = __builtin_amdgcn_rsq(h2);
            double e1 = fma(-h2 * rr, rr, 1.0);
            rr = fma(rr * e1, fma(0.375, e1, 0.5), rr);
            if (!(MODE & 2)) { e1 = fma(-h2 * rr, rr, 1.0); rr = fma(rr * 0.5, e1, rr); }
            bool const trivial = f == 0.0 || h2 < 1e-280;
            double const c = trivial ? 1.0 : fabs(d) * rr;
            double const s = trivial ? 0.0 : f * copysign(rr, d);
            double const r = trivial ? d : copysign(h2 * rr, d);
            double const yf = s * x + c * y, yc = c * x - s * y;
            if (MODE & 4) {             // lean: no selects -- lanes past the pivot compute and store values nobody reads
                asm volatile("ds_write_b128 %0, %1" :: "v"(a_rot + 16u * q), "v"(v2d{c, s}) : "memory");
                asm volatile("ds_write_b64 %0, %1" :: "v"(a_tile + 8u * (64u * (q + 1) + lane)), "v"(yf) : "memory");
                asm volatile("ds_write_b32 %0, %1" :: "v"(a_ctr), "v"(64 - q) : "memory");
                y = yc;
            } else {
            double const res = (lane < q) ? yf : r;
            asm volatile("ds_write_b128 %0, %1" :: "v"(lane == 0 ? a_rot + 16u * q : a_sink + 16u * lane), "v"(v2d{c, s}) : "memory");
            asm volatile("ds_write_b64 %0, %1" :: "v"(lane <= q ? a_tile + 8u * (64u * (q + 1) + lane) : a_sink + 8u * lane), "v"(res) : "memory");
            asm volatile("ds_write_b32 %0, %1" :: "v"(a_ctr), "v"(64 - q) : "memory");
            y = (lane < q) ? yc : (lane == q ? 0.5 : y);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    long long t1 = clock64(), w1 = wall_clock64();
    out[lane] = y;
    if (lane == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int MODE>
int run_chain_asm(double *d, long long *c, const char *what)
{
    long long h[2];
    hipLaunchKernelGGL(k_chain_asm<MODE>, 1, 64, 0, 0, d, c, 200);
    CK(hipMemcpy(h, c, 16, hipMemcpyDeviceToHost));
    printf("asm chain loop mode %d (%s): %.1f ns per rotation\n", MODE, what, h[1] * 10.0 / (200.0 * 64));
    return 0;
}

template <int MODE>
int run_chain(double *d, long long *c, const char *what)
{
    long long h[2];
    hipLaunchKernelGGL(k_chain<MODE>, 1, (MODE & 32) ? 256 : 64, 0, 0, d, c, 200);
    CK(hipMemcpy(h, c, 16, hipMemcpyDeviceToHost));
    printf("chain loop mode %2d (%s): %.1f ns per rotation\n", MODE, what, h[1] * 10.0 / (200.0 * 64));
    return 0;
}

int main()
{
    double *d; long long *c; long long h[2];
    CK(hipMalloc(&d, 64 * 8)); CK(hipMalloc(&c, 16));
    CK(hipMemset(d, 0, 64 * 8));
    int wall_khz = 0;
    CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_fma, 1, 64, 0, 0, d, c, 20000, 0.999, 0.001);
        CK(hipMemcpy(h, c, 16, hipMemcpyDeviceToHost));
        double ns = h[1] * 1e6 / wall_khz;
        printf("dependent fma_f64 : %.2f clock64 ticks, %.2f ns each (wall clock %d kHz)\n", h[0] / 320000.0, ns / 320000.0, wall_khz);
        hipLaunchKernelGGL(k_rsq, 1, 64, 0, 0, d, c, 20000, 0.5);
        CK(hipMemcpy(h, c, 16, hipMemcpyDeviceToHost));
        ns = h[1] * 1e6 / wall_khz;
        printf("dependent rsq+add : %.2f ticks, %.2f ns each\n", h[0] / 320000.0, ns / 320000.0);
        hipLaunchKernelGGL(k_step, 1, 64, 0, 0, d, c, 5000);
        CK(hipMemcpy(h, c, 16, hipMemcpyDeviceToHost));
        ns = h[1] * 1e6 / wall_khz;
        printf("chain step (regs) : %.2f ticks, %.2f ns each\n", h[0] / 160000.0, ns / 160000.0);
    }
    run_chain<0>(d, c, "arithmetic + lane read");
    run_chain<1>(d, c, "+ LDS column reads");
    run_chain<3>(d, c, "+ LDS result writes");
    run_chain<7>(d, c, "+ rotation table");
    run_chain<15>(d, c, "+ counter");
    run_chain<31>(d, c, "+ safe-range branch");
    run_chain<63>(d, c, "+ three polling waves");
    run_chain<64 + 3>(d, c, "reads + branch-free result writes");
    run_chain<64 + 7>(d, c, "reads + branch-free result writes + rotation table");
    run_chain<64 + 6>(d, c, "no reads, branch-free writes + table");
    run_chain<64 + 4>(d, c, "table only, branch-free");
    run_chain<64 + 2>(d, c, "result writes only, branch-free");
    {
        double *e; CK(hipMalloc(&e, 256 * 3 * 8));
        hipLaunchKernelGGL(k_rsq_err, 1, 256, 0, 0, e);
        double h[768]; CK(hipMemcpy(h, e, sizeof h, hipMemcpyDeviceToHost));
        double m0 = 0, m1 = 0, m2 = 0;
        for (int i = 0; i < 256; i++) { m0 = fmax(m0, h[3 * i]); m1 = fmax(m1, h[3 * i + 1]); m2 = fmax(m2, h[3 * i + 2]); }
        printf("v_rsq_f64 relative error: raw %.3g, after the cubic step %.3g, after one more %.3g (u = 1.1e-16)\n", m0, m1, m2);
    }
    run_chain_asm<1>(d, c, "production loop");
    run_chain_asm<0>(d, c, "constant fill-in");
    run_chain_asm<3>(d, c, "one Newton step");
    run_chain_asm<5>(d, c, "lean: no selects on stores and carry");
    run_chain_asm<7>(d, c, "lean + one Newton step");
    return 0;
}
