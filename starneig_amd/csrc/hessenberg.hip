// Blocked Householder Hessenberg reduction on one MI355X.
//
// Rebuilds rows H0-H10 of SURVEY.md section 8a: the panel/column/update order of
// reference hessenberg/core.c:399-596 (+ delayed updates :301-349) with the
// arithmetic of hessenberg/cpu.c:50-560, as a static schedule of HIP kernels
// on one device-resident column-major matrix (no tiles, no task graph).
//
// Per panel column j (global pivot row piv = i+1+j) the dependent chain is
//   colA  : finish Y(:,j-1) from the gemv partials, p' = P(:,j) - Y V(piv-1,:)^T,
//           partial w = V^T p'                       (cpu.c:98-115, :253-270)
//   colB  : w <- T^T * sum(partials)                 (cpu.c:118-120)
//   colC  : p'' = p' - V w, partial ||p''(piv+1:)||^2, partial V^T p''
//                                                    (cpu.c:123-130, :263-264)
//   colD  : dlarfg scalars, w_v = V^T v, T(0:j,j) = -tau T w_v (cpu.c:137-160, :277-284)
//   gemv  : y = A(i+1:end, piv:end) v  -- THE HBM-bound kernel (cpu.c:217-219,
//           cuda.cu:62-107): every trailing element is streamed once per column.
// All panel vectors/matrices (P,V,Y) are indexed by GLOBAL row so that the
// 16-byte row pairs of the gemv stay aligned for every panel offset.
#include "common.h"
#include <vector>
#include <algorithm>
#include <cmath>

namespace sn {

typedef double d2 __attribute__((ext_vector_type(2)));

// ---- 16-lane (DPP row) all-reduce of a double --------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double x)
{
    x += dpp_f64<0x128>(x);   // row_ror:8
    x += dpp_f64<0x124>(x);   // row_ror:4
    x += dpp_f64<0x122>(x);   // row_ror:2
    x += dpp_f64<0x121>(x);   // row_ror:1
    return x;
}
__device__ __forceinline__ double wave_sum(double x)
{
    x = row16_sum(x);
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
}

constexpr int RB = 256;          // rows per workgroup in the row-parallel column kernels
constexpr int GEMV_ROWS = 512;   // rows per workgroup of the big gemv (4 waves x 64 lanes x 2)
constexpr int MAX_SPLIT = 32;

// In-block transposed gemv: out[l] = sum_{r<RB} V[g0+r, l] * sp[r], l < ncols.
// 256 threads = 16 row lanes x 16 column groups; 16 lanes read 128 contiguous
// bytes of one V column, the 16-lane DPP row reduces them.
__device__ __forceinline__ void block_gemv_t(double const *__restrict__ V, int ldv,
    int g0, int ncols, double const *sp, double *__restrict__ out)
{
    int const rsub = threadIdx.x & 15, csub = threadIdx.x >> 4;
    for (int l = csub; l < ncols; l += 16) {
        double const *col = V + (size_t)l * ldv + g0 + rsub;
        double acc = 0.0;
        #pragma unroll 8
        for (int it = 0; it < RB / 16; it++)
            acc += col[it * 16] * sp[it * 16 + rsub];
        acc = row16_sum(acc);
        if (rsub == 0) out[l] = acc;
    }
}

// colA(j), j >= 1.  Row block g0..g0+RB (global rows, clipped to [R0,E)).
__global__ __launch_bounds__(256)
void hess_colA_kernel(int R0, int E, int j, int ldp,
    double *__restrict__ P, double const *__restrict__ V, double *__restrict__ Y,
    double const *__restrict__ ypart, int nsplit,
    double const *__restrict__ wv,      // j-1 entries (for column j-1)
    double const *__restrict__ scal,    // scal of column j-1: [scale, tau, beta]
    double *__restrict__ wpart, int ldw)
{
    __shared__ double s_wv[512], s_vrow[512], s_p[RB];
    int const tid = threadIdx.x;
    int const g0 = R0 + blockIdx.x * RB;
    int const g = g0 + tid;
    int const pivprev = R0 + j - 1;
    for (int l = tid; l < j; l += 256) {
        s_wv[l] = (l < j - 1) ? wv[l] : 0.0;
        s_vrow[l] = V[(size_t)l * ldp + pivprev];
    }
    __syncthreads();
    double pval = 0.0;
    if (g < E) {
        double const tau = scal[1], beta = scal[2];
        double ysum = 0.0;
        for (int s = 0; s < nsplit; s++) ysum += ypart[(size_t)s * ldp + g];
        double yacc = 0.0, pacc = 0.0;
        double const *yrow = Y + g;
        int l = 0;
        for (; l + 4 <= j - 1; l += 4) {
            double y0 = yrow[(size_t)(l + 0) * ldp], y1 = yrow[(size_t)(l + 1) * ldp];
            double y2 = yrow[(size_t)(l + 2) * ldp], y3 = yrow[(size_t)(l + 3) * ldp];
            yacc += y0 * s_wv[l] + y1 * s_wv[l + 1] + y2 * s_wv[l + 2] + y3 * s_wv[l + 3];
            pacc += y0 * s_vrow[l] + y1 * s_vrow[l + 1] + y2 * s_vrow[l + 2] + y3 * s_vrow[l + 3];
        }
        for (; l < j - 1; l++) {
            double y0 = yrow[(size_t)l * ldp];
            yacc += y0 * s_wv[l];
            pacc += y0 * s_vrow[l];
        }
        double ynew = tau * (ysum - yacc);                 // cpu.c:267-270
        Y[(size_t)(j - 1) * ldp + g] = ynew;
        pacc += ynew * s_vrow[j - 1];
        pval = P[(size_t)j * ldp + g] - pacc;              // cpu.c:98-99
        P[(size_t)j * ldp + g] = pval;
        // column j-1 of P becomes final: beta on the sub-diagonal, zeros below (cpu.c:153-154)
        if (g == pivprev) P[(size_t)(j - 1) * ldp + g] = beta;
        else if (g > pivprev) P[(size_t)(j - 1) * ldp + g] = 0.0;
    }
    s_p[tid] = pval;
    __syncthreads();
    // rows past E contribute zero through s_p; V reads stay inside the padded buffer
    block_gemv_t(V, ldp, g0, j, s_p, wpart + (size_t)blockIdx.x * ldw);
}

// Last column of a panel: only finish Y(:,nb-1) and finalize P(:,nb-1).
__global__ __launch_bounds__(256)
void hess_finish_kernel(int R0, int E, int j /* = nb */, int ldp,
    double *__restrict__ P, double *__restrict__ Y,
    double const *__restrict__ ypart, int nsplit,
    double const *__restrict__ wv, double const *__restrict__ scal)
{
    __shared__ double s_wv[512];
    int const tid = threadIdx.x;
    int const g = R0 + blockIdx.x * RB + tid;
    int const pivprev = R0 + j - 1;
    for (int l = tid; l < j - 1; l += 256) s_wv[l] = wv[l];
    __syncthreads();
    if (g < E) {
        double const tau = scal[1], beta = scal[2];
        double ysum = 0.0;
        for (int s = 0; s < nsplit; s++) ysum += ypart[(size_t)s * ldp + g];
        double yacc = 0.0;
        for (int l = 0; l < j - 1; l++) yacc += Y[(size_t)l * ldp + g] * s_wv[l];
        Y[(size_t)(j - 1) * ldp + g] = tau * (ysum - yacc);
        if (g == pivprev) P[(size_t)(j - 1) * ldp + g] = beta;
        else if (g > pivprev) P[(size_t)(j - 1) * ldp + g] = 0.0;
    }
}

// colB(j): w = TT * sum_wg wpart  (TT = T^T kept explicitly, lower triangular)
__global__ __launch_bounds__(512)
void hess_colB_kernel(int j, int nwg, double const *__restrict__ wpart, int ldw,
    double const *__restrict__ TT, int ldt, double *__restrict__ w)
{
    __shared__ double s_w[512];
    int const l = threadIdx.x;
    if (l < j) {
        double s = 0.0;
        for (int b = 0; b < nwg; b++) s += wpart[(size_t)b * ldw + l];
        s_w[l] = s;
    }
    __syncthreads();
    if (l < j) {
        double s = 0.0;
        for (int r = 0; r <= l; r++) s += TT[(size_t)r * ldt + l] * s_w[r];   // T(r,l) w(r)
        w[l] = s;
    }
}

// colC(j): p'' = p' - V(:,0:j) w ; partial norm^2 below the pivot ; partial V^T p''(piv+1:)
__global__ __launch_bounds__(256)
void hess_colC_kernel(int R0, int E, int j, int ldp,
    double *__restrict__ P, double const *__restrict__ V,
    double const *__restrict__ w, double *__restrict__ normpart,
    double *__restrict__ wvpart, int ldw)
{
    __shared__ double s_w[512], s_p[RB], s_red[4];
    int const tid = threadIdx.x;
    int const g0 = R0 + blockIdx.x * RB;
    int const g = g0 + tid;
    int const piv = R0 + j;
    for (int l = tid; l < j; l += 256) s_w[l] = w[l];
    __syncthreads();
    double pval = 0.0;
    if (g < E) {
        double acc = 0.0;
        double const *vrow = V + g;
        int l = 0;
        for (; l + 4 <= j; l += 4)
            acc += vrow[(size_t)(l + 0) * ldp] * s_w[l] + vrow[(size_t)(l + 1) * ldp] * s_w[l + 1]
                 + vrow[(size_t)(l + 2) * ldp] * s_w[l + 2] + vrow[(size_t)(l + 3) * ldp] * s_w[l + 3];
        for (; l < j; l++) acc += vrow[(size_t)l * ldp] * s_w[l];
        pval = P[(size_t)j * ldp + g];
        if (j > 0) { pval -= acc; P[(size_t)j * ldp + g] = pval; }   // cpu.c:123-130
    }
    double below = (g > piv && g < E) ? pval : 0.0;
    s_p[tid] = below;
    double ss = wave_sum(below * below);
    if ((tid & 63) == 0) s_red[tid >> 6] = ss;
    __syncthreads();
    if (tid == 0) normpart[blockIdx.x] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    block_gemv_t(V, ldp, g0, j, s_p, wvpart + (size_t)blockIdx.x * ldw);
}

// colD(j): reflector scalars (LAPACK dlarfg, cpu.c:137-141), w_v = V(piv:,0:j)^T v,
// T(0:j,j) = -tau T(0:j,0:j) w_v, T(j,j) = tau (cpu.c:277-284).  One workgroup.
__global__ __launch_bounds__(512)
void hess_colD_kernel(int R0, int j, int ldp, int nwg,
    double const *__restrict__ P, double const *__restrict__ V,
    double const *__restrict__ normpart, double const *__restrict__ wvpart, int ldw,
    double *__restrict__ T, double *__restrict__ TT, int ldt,
    double *__restrict__ wv, double *__restrict__ scal)
{
    __shared__ double s_wv[512], s_red[8], s_scal[3];
    int const tid = threadIdx.x;
    int const piv = R0 + j;
    double part = 0.0;
    for (int b = tid; b < nwg; b += 512) part += normpart[b];
    part = wave_sum(part);
    if ((tid & 63) == 0) s_red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        double ssq = 0.0;
        for (int k = 0; k < 8; k++) ssq += s_red[k];
        double alpha = P[(size_t)j * ldp + piv];
        double xnorm = sqrt(ssq);
        double tau = 0.0, scale = 0.0, beta = alpha;
        if (xnorm != 0.0) {
            beta = -copysign(hypot(alpha, xnorm), alpha);
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        s_scal[0] = scale; s_scal[1] = tau; s_scal[2] = beta;
        scal[0] = scale; scal[1] = tau; scal[2] = beta;
    }
    __syncthreads();
    double const scale = s_scal[0], tau = s_scal[1];
    if (tid < j) {
        double s = 0.0;
        for (int b = 0; b < nwg; b++) s += wvpart[(size_t)b * ldw + tid];
        double x = V[(size_t)tid * ldp + piv] + scale * s;     // v(piv) = 1
        s_wv[tid] = x;
        wv[tid] = x;
    }
    __syncthreads();
    if (tid < j) {
        double s = 0.0;
        for (int r = tid; r < j; r++) s += T[(size_t)r * ldt + tid] * s_wv[r];
        s *= -tau;
        T[(size_t)j * ldt + tid] = s;
        TT[(size_t)tid * ldt + j] = s;
    }
    if (tid == 0) { T[(size_t)j * ldt + j] = tau; TT[(size_t)j * ldt + j] = tau; }
}

// The big gemv: ypart[split][g] = sum_{c in split} A[g, c] * v[c],  g in [R0,E),
// c in [piv,E), v[piv] = 1, v[c] = scale * p''[c].  Also materialises V(:,j).
// Workgroup = 512 rows (each lane owns an aligned row pair, 16-byte loads) x one
// column chunk; the 4 waves of a workgroup read 4 KiB contiguous per column.
template <int UNROLL>
__global__ __launch_bounds__(256)
void hess_gemv_kernel(double const *__restrict__ A, int ldA,
    double const *__restrict__ pcol, double const *__restrict__ scal,
    int R0, int E, int piv, int cols_per_split, int ldp,
    double *__restrict__ ypart, double *__restrict__ Vcol)
{
    int const g = (R0 & ~1) + blockIdx.x * GEMV_ROWS + threadIdx.x * 2;
    int const c_begin = piv + blockIdx.y * cols_per_split;
    int const c_end = min(E, c_begin + cols_per_split);
    double const scale = scal[0];
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    if (g < E) {
        double const *a = A + (size_t)c_begin * ldA + g;
        int c = c_begin;
        for (; c + UNROLL <= c_end; c += UNROLL) {
            d2 x[UNROLL];
            #pragma unroll
            for (int u = 0; u < UNROLL; u++)
                x[u] = *reinterpret_cast<d2 const *>(a + (size_t)u * ldA);
            #pragma unroll
            for (int u = 0; u < UNROLL; u += 2) {
                double v0 = (c + u == piv) ? 1.0 : scale * pcol[c + u];
                double v1 = scale * pcol[c + u + 1];
                a0 += x[u].x * v0;     a1 += x[u].y * v0;
                b0 += x[u + 1].x * v1; b1 += x[u + 1].y * v1;
            }
            a += (size_t)UNROLL * ldA;
        }
        for (; c < c_end; c++) {
            d2 x = *reinterpret_cast<d2 const *>(a);
            double v0 = (c == piv) ? 1.0 : scale * pcol[c];
            a0 += x.x * v0; a1 += x.y * v0;
            a += ldA;
        }
        double *yp = ypart + (size_t)blockIdx.y * ldp;
        if (g >= R0) yp[g] = a0 + b0;
        if (g + 1 < E) yp[g + 1] = a1 + b1;
        if (blockIdx.y == 0) {
            #pragma unroll
            for (int q = 0; q < 2; q++) {
                int gg = g + q;
                if (gg >= R0 && gg < E)
                    Vcol[gg] = gg < piv ? 0.0 : (gg == piv ? 1.0 : scale * pcol[gg]);
            }
        }
    }
}

// Fallback for odd leading dimensions / unaligned bases (8-byte loads).
__global__ __launch_bounds__(256)
void hess_gemv_unaligned_kernel(double const *__restrict__ A, int ldA,
    double const *__restrict__ pcol, double const *__restrict__ scal,
    int R0, int E, int piv, int cols_per_split, int ldp,
    double *__restrict__ ypart, double *__restrict__ Vcol)
{
    int const c_begin = piv + blockIdx.y * cols_per_split;
    int const c_end = min(E, c_begin + cols_per_split);
    double const scale = scal[0];
    for (int q = 0; q < 2; q++) {
        int const g = R0 + blockIdx.x * GEMV_ROWS + q * 256 + threadIdx.x;
        if (g >= E) continue;
        double acc = 0.0;
        double const *a = A + (size_t)c_begin * ldA + g;
        for (int c = c_begin; c < c_end; c++, a += ldA)
            acc += (*a) * ((c == piv) ? 1.0 : scale * pcol[c]);
        ypart[(size_t)blockIdx.y * ldp + g] = acc;
        if (blockIdx.y == 0)
            Vcol[g] = g < piv ? 0.0 : (g == piv ? 1.0 : scale * pcol[g]);
    }
}

__global__ void hess_copy_in_kernel(int R0, int E, int nb, int i,
    double const *__restrict__ A, int ldA, double *__restrict__ P, int ldp)
{
    int g = R0 + blockIdx.x * 256 + threadIdx.x;
    int j = blockIdx.y;
    if (g < E && j < nb) P[(size_t)j * ldp + g] = A[(size_t)(i + j) * ldA + g];
}
__global__ void hess_copy_out_kernel(int R0, int E, int nb, int i,
    double *__restrict__ A, int ldA, double const *__restrict__ P, int ldp)
{
    int g = R0 + blockIdx.x * 256 + threadIdx.x;
    int j = blockIdx.y;
    if (g < E && j < nb) A[(size_t)(i + j) * ldA + g] = P[(size_t)j * ldp + g];
}

// ---- workspace ----------------------------------------------------------------
struct HessWorkspace {
    int n = 0, nbmax = 0, ldp = 0, ldw = 0, nwg_max = 0;
    double *P = nullptr, *V[2] = {nullptr, nullptr}, *Y = nullptr, *VT[2] = {nullptr, nullptr};
    double *T[2] = {nullptr, nullptr}, *TT = nullptr;
    double *W = nullptr, *W2 = nullptr;
    double *ypart = nullptr, *wpart = nullptr, *wvpart = nullptr, *normpart = nullptr;
    double *w = nullptr, *wv = nullptr, *scal = nullptr;
    hipStream_t side = nullptr;
    hipEvent_t panel_done[2] = {nullptr, nullptr}, side_done[2] = {nullptr, nullptr};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> sample_ev;      // pairs (start, stop) around sampled gemv launches
    std::vector<double> sample_bytes;

    void release() {
        double **ptrs[] = {&P, &V[0], &V[1], &Y, &VT[0], &VT[1], &T[0], &T[1], &TT, &W, &W2,
            &ypart, &wpart, &wvpart, &normpart, &w, &wv, &scal};
        for (auto p : ptrs) if (*p) { SN_HIP_CHECK(hipFree(*p)); *p = nullptr; }
        n = nbmax = 0;
    }
    void ensure(int n_, int nb_) {
        if (n_ <= n && nb_ <= nbmax) return;
        release();
        n = n_; nbmax = nb_;
        ldp = (int)roundup((size_t)n + RB + 16, 128);
        ldw = (int)roundup((size_t)nbmax, 16);
        nwg_max = divceil(n, RB) + 1;
        size_t pan = (size_t)ldp * nbmax * sizeof(double);
        auto alloc = [](double **p, size_t bytes) {
            SN_HIP_CHECK(hipMalloc((void **)p, bytes));
            SN_HIP_CHECK(hipMemset(*p, 0, bytes));
        };
        alloc(&P, pan); alloc(&V[0], pan); alloc(&V[1], pan); alloc(&Y, pan);
        alloc(&VT[0], pan); alloc(&VT[1], pan);
        size_t tb = (size_t)nbmax * nbmax * sizeof(double);
        alloc(&T[0], tb); alloc(&T[1], tb); alloc(&TT, tb);
        alloc(&W, pan); alloc(&W2, pan);
        alloc(&ypart, (size_t)MAX_SPLIT * ldp * sizeof(double));
        alloc(&wpart, (size_t)nwg_max * ldw * sizeof(double));
        alloc(&wvpart, (size_t)nwg_max * ldw * sizeof(double));
        alloc(&normpart, (size_t)nwg_max * sizeof(double));
        alloc(&w, 512 * sizeof(double)); alloc(&wv, 512 * sizeof(double));
        alloc(&scal, 16 * sizeof(double));
        if (!side) {
            SN_HIP_CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
            for (int k = 0; k < 2; k++) {
                SN_HIP_CHECK(hipEventCreateWithFlags(&panel_done[k], hipEventDisableTiming));
                SN_HIP_CHECK(hipEventCreateWithFlags(&side_done[k], hipEventDisableTiming));
            }
            SN_HIP_CHECK(hipEventCreate(&ev0));
            SN_HIP_CHECK(hipEventCreate(&ev1));
        }
    }
};

static HessWorkspace g_ws;

void hessenberg_release_workspace() { g_ws.release(); }

static void choose_split(int m_rows, int ncols, int *nsplit, int *cps)
{
    int row_tiles = divceil(m_rows + 1, GEMV_ROWS);
    int want = std::max(1, 1536 / row_tiles);           // ~6 workgroups per CU
    int s = std::min({want, MAX_SPLIT, std::max(1, ncols / 16)});
    int c = divceil(ncols, s);
    c = (c + 7) / 8 * 8;
    s = divceil(ncols, c);
    *nsplit = std::max(1, s);
    *cps = c;
}

int hessenberg_device(hipStream_t s, int n, int begin, int end, int panel_width,
    double *dA, int ldA, double *dQ, int ldQ, HessenbergTimings *tm)
{
    if (panel_width > 504) panel_width = 504;           // column kernels hold j <= 512 in LDS
    HessWorkspace &ws = g_ws;
    ws.ensure(n, panel_width);
    int const ldp = ws.ldp, ldw = ws.ldw;
    bool const aligned = (ldA % 2 == 0) && (((uintptr_t)dA) % 16 == 0);
    double gemv_bytes = 0.0, gemm_flops = 0.0;
    long gemv_launches = 0;
    size_t nsampled = 0;
    int const sample_every = tm ? tm->sample_every : 0;
    ws.sample_bytes.clear();

    if (tm) SN_HIP_CHECK(hipEventRecord(ws.ev0, s));
    int pcount = 0;
    for (int i = begin; i < end - 1; i += panel_width, pcount++) {
        int const nb = std::min(panel_width, end - i - 1);
        int const R0 = i + 1, E = end, m = E - R0;
        int const buf = pcount & 1;
        double *V = ws.V[buf], *VT = ws.VT[buf], *T = ws.T[buf];
        int const nwg = divceil(m, RB);

        // V/VT/T of this slot were last used by the side stream two panels ago
        if (pcount >= 2) SN_HIP_CHECK(hipStreamWaitEvent(s, ws.side_done[buf], 0));

        SN_HIP_CHECK(hipMemsetAsync(T, 0, (size_t)ws.nbmax * ws.nbmax * sizeof(double), s));
        SN_HIP_CHECK(hipMemsetAsync(ws.TT, 0, (size_t)ws.nbmax * ws.nbmax * sizeof(double), s));
        hipLaunchKernelGGL(hess_copy_in_kernel, dim3(divceil(m, 256), nb), dim3(256), 0, s,
            R0, E, nb, i, dA, ldA, ws.P, ldp);                                // core.c:451

        int nsplit = 1, cps = 0;
        for (int j = 0; j < nb; j++) {
            int const piv = R0 + j;
            if (j > 0) {
                hipLaunchKernelGGL(hess_colA_kernel, dim3(nwg), dim3(256), 0, s,
                    R0, E, j, ldp, ws.P, V, ws.Y, ws.ypart, nsplit, ws.wv, ws.scal,
                    ws.wpart, ldw);
                hipLaunchKernelGGL(hess_colB_kernel, dim3(1), dim3(512), 0, s,
                    j, nwg, ws.wpart, ldw, ws.TT, ws.nbmax, ws.w);
            }
            hipLaunchKernelGGL(hess_colC_kernel, dim3(nwg), dim3(256), 0, s,
                R0, E, j, ldp, ws.P, V, ws.w, ws.normpart, ws.wvpart, ldw);
            hipLaunchKernelGGL(hess_colD_kernel, dim3(1), dim3(512), 0, s,
                R0, j, ldp, nwg, ws.P, V, ws.normpart, ws.wvpart, ldw,
                T, ws.TT, ws.nbmax, ws.wv, ws.scal);
            int const ncols = E - piv;
            choose_split(m, ncols, &nsplit, &cps);
            dim3 grid(divceil(E - (R0 & ~1), GEMV_ROWS), nsplit);
            bool const sampled = sample_every > 0 && (gemv_launches % sample_every) == 0;
            if (sampled) {
                if (ws.sample_ev.size() < 2 * (nsampled + 1)) {
                    hipEvent_t a, b;
                    SN_HIP_CHECK(hipEventCreate(&a));
                    SN_HIP_CHECK(hipEventCreate(&b));
                    ws.sample_ev.push_back(a); ws.sample_ev.push_back(b);
                }
                SN_HIP_CHECK(hipEventRecord(ws.sample_ev[2 * nsampled], s));
            }
            if (aligned)
                hipLaunchKernelGGL(hess_gemv_kernel<8>, grid, dim3(256), 0, s,
                    dA, ldA, ws.P + (size_t)j * ldp, ws.scal, R0, E, piv, cps, ldp,
                    ws.ypart, V + (size_t)j * ldp);
            else
                hipLaunchKernelGGL(hess_gemv_unaligned_kernel,
                    dim3(divceil(m, GEMV_ROWS), nsplit), dim3(256), 0, s,
                    dA, ldA, ws.P + (size_t)j * ldp, ws.scal, R0, E, piv, cps, ldp,
                    ws.ypart, V + (size_t)j * ldp);
            if (sampled) {
                SN_HIP_CHECK(hipEventRecord(ws.sample_ev[2 * nsampled + 1], s));
                ws.sample_bytes.push_back(8.0 * (double)m * (double)ncols);
                nsampled++;
            }
            gemv_launches++;
            gemv_bytes += 8.0 * (double)m * (double)ncols;
        }
        hipLaunchKernelGGL(hess_finish_kernel, dim3(nwg), dim3(256), 0, s,
            R0, E, nb, ldp, ws.P, ws.Y, ws.ypart, nsplit, ws.wv, ws.scal);

        // VT = V * T  (so that every W = X^T/X * V * T below is one GEMM)
        dgemm(s, 'N', 'N', m, nb, nb, 1.0, V + R0, ldp, T, ws.nbmax, 0.0, VT + R0, ldp);
        gemm_flops += 2.0 * m * nb * nb;

        // ---- critical trailing updates (core.c:523-547) ----
        int const nt = E - (i + nb);
        if (nt > 0) {
            double *At = dA + (size_t)(i + nb) * ldA + R0;
            dgemm(s, 'N', 'T', m, nt, nb, -1.0, ws.Y + R0, ldp, V + (i + nb), ldp, 1.0, At, ldA);
            dgemm(s, 'T', 'N', nt, nb, m, 1.0, At, ldA, VT + R0, ldp, 0.0, ws.W, ldp);
            dgemm(s, 'N', 'T', m, nt, nb, -1.0, V + R0, ldp, ws.W, ldp, 1.0, At, ldA);
            gemm_flops += 6.0 * m * (double)nt * nb;
        }
        // panel columns go back into A (core.c:317)
        hipLaunchKernelGGL(hess_copy_out_kernel, dim3(divceil(m, 256), nb), dim3(256), 0, s,
            R0, E, nb, i, dA, ldA, ws.P, ldp);
        SN_HIP_CHECK(hipEventRecord(ws.panel_done[buf], s));

        // ---- non-critical updates on the side stream (core.c:321-340) ----
        hipStream_t q = ws.side;
        SN_HIP_CHECK(hipStreamWaitEvent(q, ws.panel_done[buf], 0));
        {   // upper rows A(0:R0, R0:E) (I - V T V^T)
            double *X = dA + (size_t)R0 * ldA;
            dgemm(q, 'N', 'N', R0, nb, m, 1.0, X, ldA, VT + R0, ldp, 0.0, ws.W2, ldp);
            dgemm(q, 'N', 'T', R0, m, nb, -1.0, ws.W2, ldp, V + R0, ldp, 1.0, X, ldA);
            gemm_flops += 4.0 * R0 * (double)m * nb;
        }
        if (E < n) {   // columns right of end (core.c:330-336)
            int const nr = n - E;
            double *At2 = dA + (size_t)E * ldA + R0;
            dgemm(q, 'T', 'N', nr, nb, m, 1.0, At2, ldA, VT + R0, ldp, 0.0, ws.W2, ldp);
            dgemm(q, 'N', 'T', m, nr, nb, -1.0, V + R0, ldp, ws.W2, ldp, 1.0, At2, ldA);
            gemm_flops += 4.0 * nr * (double)m * nb;
        }
        if (dQ) {   // Q(:, R0:E) (I - V T V^T)
            double *X = dQ + (size_t)R0 * ldQ;
            dgemm(q, 'N', 'N', n, nb, m, 1.0, X, ldQ, VT + R0, ldp, 0.0, ws.W2, ldp);
            dgemm(q, 'N', 'T', n, m, nb, -1.0, ws.W2, ldp, V + R0, ldp, 1.0, X, ldQ);
            gemm_flops += 4.0 * n * (double)m * nb;
        }
        SN_HIP_CHECK(hipEventRecord(ws.side_done[buf], q));
    }
    // join the side stream back into s
    for (int k = 0; k < 2 && k < pcount; k++)
        SN_HIP_CHECK(hipStreamWaitEvent(s, ws.side_done[k], 0));
    if (tm) {
        SN_HIP_CHECK(hipEventRecord(ws.ev1, s));
        SN_HIP_CHECK(hipEventSynchronize(ws.ev1));
        float ms = 0.f;
        SN_HIP_CHECK(hipEventElapsedTime(&ms, ws.ev0, ws.ev1));
        tm->total_ms = ms;
        tm->gemv_bytes = gemv_bytes;
        tm->gemm_flops = gemm_flops;
        tm->gemv_launches = gemv_launches;
        tm->sampled_launches = (long)nsampled;
        tm->sampled_bytes = 0.0; tm->sampled_ms = 0.0;
        for (size_t k = 0; k < nsampled; k++) {
            float t = 0.f;
            SN_HIP_CHECK(hipEventElapsedTime(&t, ws.sample_ev[2 * k], ws.sample_ev[2 * k + 1]));
            tm->sampled_ms += t;
            tm->sampled_bytes += ws.sample_bytes[k];
        }
    }
    return 0;
}

} // namespace sn
