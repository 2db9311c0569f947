// Scratch: where does a column of group_qr (starneig_amd/csrc/ht_twostage.hip) spend its cycles?  One workgroup,
// 64 x 63, phases timed with s_memtime on the publishing group and on the last group.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
constexpr int QT = 1024;
template <int CTRL>
__device__ __forceinline__ double ht2_dpp(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double x)
{
    x += ht2_dpp<0x128>(x); x += ht2_dpp<0x124>(x); x += ht2_dpp<0x122>(x); x += ht2_dpp<0x121>(x);
    return x;
}
#define CLK() __builtin_readcyclecounter()
template <int MR, int MODE>
__global__ __launch_bounds__(QT) void qr_kernel(double *Pg, int ldp, int m, int k, long long *out)
{
    extern __shared__ double P[];
    __shared__ double piv[2][2];
    __shared__ double tau[64], scl[64];
    int const tid = threadIdx.x, l = tid & 15, g = tid >> 4;
    for (int e = tid; e < ldp * k; e += QT) P[e] = Pg[e];
    __syncthreads();
    int const kref = min(m - 1, k);
    double x[MR];
    #pragma unroll
    for (int r = 0; r < MR; r++) { int const i = l + 16 * r; x[r] = (g < k && i < m) ? P[g * ldp + i] : 0.0; }
    long long c_red = 0, c_math = 0, c_store = 0, c_upd = 0, c_bar = 0, c_pivread = 0;
    auto publish = [&](int c) {
        long long t0 = CLK();
        double s = 0.0, alpha = 0.0;
        #pragma unroll
        for (int r = 0; r < MR; r++) { int const i = l + 16 * r; if (i > c) s += x[r] * x[r]; if (i == c) alpha = x[r]; }
        s = row16_sum(s); alpha = row16_sum(alpha);
        long long t1 = CLK();
        double t = 0.0, beta = alpha, scale = 0.0;
        if (MODE == 0) {
            if (s != 0.0) { beta = -copysign(sqrt(alpha * alpha + s), alpha); t = (beta - alpha) / beta; scale = 1.0 / (alpha - beta); }
        } else {
            if (s != 0.0) {
                double const nn = alpha * alpha + s;
                double rs = __builtin_amdgcn_rsq(nn);
                rs = rs * (1.5 - 0.5 * nn * rs * rs);            // one Newton step
                rs = rs * (1.5 - 0.5 * nn * rs * rs);
                double const nrm = nn * rs;
                beta = -copysign(nrm, alpha);
                double const rb = -copysign(rs, alpha);          // 1 / beta
                t = (beta - alpha) * rb;
                double const dd = alpha - beta;
                double rc = __builtin_amdgcn_rcp(dd);
                rc = rc * (2.0 - dd * rc); rc = rc * (2.0 - dd * rc);
                scale = rc;
            }
        }
        long long t2 = CLK();
        #pragma unroll
        for (int r = 0; r < MR; r++) { int const i = l + 16 * r; if (i < m) P[c * ldp + i] = (i == c) ? beta : x[r]; }
        if (l == 0) { tau[c] = t; scl[c] = scale; piv[c & 1][0] = t; piv[c & 1][1] = scale; }
        long long t3 = CLK();
        c_red += t1 - t0; c_math += t2 - t1; c_store += t3 - t2;
    };
    if (g == 0 && kref > 0) publish(0);
    __syncthreads();
    long long tstart = CLK();
    for (int c = 0; c < kref; c++) {
        long long t0 = CLK();
        double const t = piv[c & 1][0], scale = piv[c & 1][1];
        long long t1 = CLK();
        c_pivread += t1 - t0;
        if (g > c && g < k) {
            if (t != 0.0) {
                double const *col = P + c * ldp;
                double pc[MR], w = 0.0, xc = 0.0;
                #pragma unroll
                for (int r = 0; r < MR; r++) {
                    int const i = l + 16 * r;
                    pc[r] = (i > c && i < m) ? col[i] : 0.0;
                    w += pc[r] * x[r];
                    if (i == c) xc = x[r];
                }
                w = row16_sum(w); xc = row16_sum(xc);
                w = (w * scale + xc) * t;
                double const wsc = w * scale;
                #pragma unroll
                for (int r = 0; r < MR; r++) { int const i = l + 16 * r; x[r] -= wsc * pc[r]; if (i == c) x[r] -= w; }
            }
            long long t2 = CLK();
            c_upd += t2 - t1;
            if (g == c + 1 && c + 1 < kref) publish(c + 1);
        }
        long long t3 = CLK();
        __syncthreads();
        c_bar += CLK() - t3;
    }
    long long total = CLK() - tstart;
    if (g >= kref && g < k) {
        #pragma unroll
        for (int r = 0; r < MR; r++) { int const i = l + 16 * r; if (i < m) P[g * ldp + i] = x[r]; }
    }
    __syncthreads();
    for (int e = tid; e < ldp * k; e += QT) Pg[e] = P[e];
    if (l == 0) {
        long long *o = out + g * 8;
        o[0] = total; o[1] = c_pivread; o[2] = c_upd; o[3] = c_red; o[4] = c_math; o[5] = c_store; o[6] = c_bar;
    }
}
template <int MR, int MODE> void run(int m, int k, const char *name)
{
    int const ldp = (MR == 4 ? 65 : 129);
    std::vector<double> h(ldp * 64);
    srand(1); for (auto &v : h) v = rand() / (double)RAND_MAX - 0.5;
    double *d; long long *o; hipMalloc(&d, h.size() * 8); hipMalloc(&o, 64 * 8 * 8);
    size_t const lds = ldp * 64 * 8;
    hipFuncSetAttribute((const void *)qr_kernel<MR, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int it = 0; it < 3; it++) {
        hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a); hipLaunchKernelGGL((qr_kernel<MR, MODE>), dim3(1), dim3(QT), lds, 0, d, ldp, m, k, o); hipEventRecord(b);
        hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<long long> ho(64 * 8); hipMemcpy(ho.data(), o, ho.size() * 8, hipMemcpyDeviceToHost);
        if (it == 2) {
            long long red = 0, math = 0, store = 0; for (int g = 0; g < 64; g++) { red += ho[g * 8 + 3]; math += ho[g * 8 + 4]; store += ho[g * 8 + 5]; }
            std::vector<double> r(h.size()); hipMemcpy(r.data(), d, r.size() * 8, hipMemcpyDeviceToHost);
            double chk = 0; for (int j = 0; j < k; j++) chk += fabs(r[j * ldp + std::min(j, m - 1)]);
            printf("%s m=%d k=%d: kernel %.1f us; loop cycles (group 63) %lld = pivot read %lld + update %lld + barrier wait %lld ...; "
                   "publishers summed: reduce %lld, sqrt/div %lld, store %lld; sum|R_jj| %.12f\n", name, m, k, ms * 1e3,
                   ho[63 * 8], ho[63 * 8 + 1], ho[63 * 8 + 2], ho[63 * 8 + 6], red, math, store, chk);
        }
    }
}
int main()
{
    run<4, 0>(64, 63, "libm sqrt/div");
    run<4, 1>(64, 63, "rsq/rcp + Newton");
    run<8, 0>(128, 64, "libm sqrt/div");
    run<8, 1>(128, 64, "rsq/rcp + Newton");
    return 0;
}
