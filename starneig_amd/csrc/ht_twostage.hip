// Two-stage reduction of a pencil (A, B), B upper triangular, to Hessenberg-triangular form (the step the
// reference delegates to LAPACK dgghd3, wrappers/lapack.c:143-163) -- the alternative to the rotation path of
// hess_tri.hip whose n^2/2 dependent column rotations bound it.
//
//   Stage 1 (Dackland & Kagstrom; Kagstrom, Kressner, Quintana-Orti, Quintana-Orti 2008): block column by block
//   column (r = 64 wide), bottom up, QR of the (2r x r) blocks of the panel -- reflectors from the left on A, B, Q;
//   the diagonal block of B they fill is restored by r reflectors from the right (RQ of its bottom r rows) on
//   B, A, Z; one RQ of the full r x r block that remains at the top of the block column.  A ends with r
//   sub-diagonals.  Compact-WY factors, every application three fp64 MFMA GEMMs.
//   Stage 2: a Householder bulge chase.  Sweep j, position t: the left reflector of length r that reduces the
//   overhanging column (rows p .. p + r - 1, p = j + 1 + t r), applied to the rows of A, B and to Q; the
//   "opposite" reflector from the right whose first column is orthogonal to rows 2 .. r of the r x r block of B
//   (QR of those rows in LDS: no solve with B, singular B included), applied to the columns of B, A and to Z.
//   Sweep j + 1 may run position t once sweep j has left position t + 2: wavefronts of ~n / (3 r) independent
//   steps, four launches each (build / apply left, build / apply right).
// scratch/ht2_proto.py is the numpy statement of the same algorithm (tests/test_ht_twostage_prototype.py).
#include "common.h"
#include "tuning.h"
#include <algorithm>
#include <cmath>
#include <vector>

namespace sn {

namespace {

constexpr int R2 = 64;              // band width of stage 1 = reflector length of stage 2
constexpr int QT = 1024;            // threads of the factorisation kernels
constexpr int LDP = 2 * R2 + 1;     // LDS leading dimension of a (2 r x r) panel (odd: conflict-free walks)

__device__ __forceinline__ double wsum64(double x)
{
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// Householder QR of the m x k matrix P in LDS (column-major, leading dimension ldp) by the whole workgroup (QT
// threads): R in the upper triangle, the reflectors' vectors below the diagonal (unit diagonal implied), tau[0:k];
// scl[c] = 1 (the vectors are scaled in place).  LAPACK dlarfg conventions.  Three workgroup barriers per column:
// 2.7 us per column, 170 us for the 64 x 63 factorisation behind an opposite reflector, 270 us for a 128 x 64
// panel with its T factor.  (A second version gave every column to ONE lane of ONE wave -- no barrier, no
// reduction on the chain -- and was slower: a single wave hides no LDS latency; profiles/r5_ht_twostage_*.txt.)
__device__ void wave_qr(double *P, int ldp, int m, int k, double *tau, double *scl)
{
    __shared__ double red[2];
    int const tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int const NW = blockDim.x / 64;
    int const kref = min(m - 1, k);
    for (int c = 0; c < kref; c++) {
        double *col = P + c * ldp;
        if (wave == 0) {
            double ss = 0.0;
            for (int i = c + 1 + lane; i < m; i += 64) { double const x = col[i]; ss += x * x; }
            ss = wsum64(ss);
            if (lane == 0) {
                double const alpha = col[c];
                double t = 0.0, beta = alpha, scale = 0.0;
                if (ss != 0.0) {
                    beta = -copysign(sqrt(alpha * alpha + ss), alpha);
                    t = (beta - alpha) / beta; scale = 1.0 / (alpha - beta);
                }
                tau[c] = t; scl[c] = 1.0; red[0] = scale; red[1] = beta;
            }
        }
        __syncthreads();
        double const scale = red[0], t = tau[c];
        for (int i = c + 1 + tid; i < m; i += blockDim.x) col[i] *= scale;
        if (tid == 0) col[c] = red[1];
        __syncthreads();
        if (t != 0.0)
            for (int j = c + 1 + wave; j < k; j += NW) {
                double *cj = P + j * ldp;
                double w = 0.0;
                for (int i = c + 1 + lane; i < m; i += 64) w += col[i] * cj[i];
                w = (wsum64(w) + cj[c]) * t;
                for (int i = c + 1 + lane; i < m; i += 64) cj[i] -= w * col[i];
                if (lane == 0) cj[c] -= w;
            }
        __syncthreads();
    }
    for (int c = kref + tid; c < k; c += blockDim.x) { tau[c] = 0.0; scl[c] = 1.0; }
    __syncthreads();
}

// T (k x k, upper triangular, leading dimension ldt, in LDS) of the compact-WY form H_0 H_1 ... H_{k-1} =
// I - V T V^T from the panel wave_qr left (v_c = [0 .. 0, 1, scl[c] P(c+1:m, c)]; LAPACK dlarft, forward /
// columnwise).  All threads of the workgroup: the Gram matrix entry by entry, then every row of T by its own
// thread (row i of T depends on row i alone).  G: k x k scratch in LDS.
__device__ void lds_tfactor(double const *P, int ldp, int m, int k, double const *tau, double const *scl,
    double *T, int ldt, double *G)
{
    int const tid = threadIdx.x, nt = blockDim.x;
    for (int e = tid; e < k * k; e += nt) {
        int const i = e % k, j = e / k;
        T[j * ldt + i] = 0.0;
        if (i >= j) continue;
        double s = 0.0;
        for (int r = j + 1; r < m; r++) s += P[i * ldp + r] * P[j * ldp + r];
        // v_i^T v_j = v_i(j) * 1 + sum_{r > j} v_i(r) v_j(r)
        G[j * k + i] = scl[i] * ((j < m ? P[i * ldp + j] : 0.0) + scl[j] * s);
    }
    __syncthreads();
    if (tid < k) {
        int const i = tid;
        T[i * ldt + i] = tau[i];
        for (int j = i + 1; j < k; j++) {
            double s = 0.0;
            for (int l = i; l < j; l++) s += T[l * ldt + i] * G[j * k + l];
            T[j * ldt + i] = -tau[j] * s;
        }
    }
    __syncthreads();
}

// Stage 1, left: QR of the m x nb block X = A(i0:i0+m, jc:jc+nb) (m <= 2 r, nb <= r); R back in place with
// exact zeros below it, V (m x nb, unit lower trapezoidal, leading dimension ldv) and T (nb x nb, ld R2) out.
__global__ __launch_bounds__(QT) void ht2_panel_qr_kernel(double *__restrict__ X, int ldx, int m, int nb,
    double *__restrict__ V, int ldv, double *__restrict__ T)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *P = lds, *Tl = P + R2 * LDP, *G = Tl + R2 * R2, *tau = G + R2 * R2, *scl = tau + R2;
    int const tid = threadIdx.x;
    for (int e = tid; e < m * nb; e += QT) { int const i = e % m, j = e / m; P[j * LDP + i] = X[(size_t)j * ldx + i]; }
    __syncthreads();
    wave_qr(P, LDP, m, nb, tau, scl);
    lds_tfactor(P, LDP, m, nb, tau, scl, Tl, R2, G);
    for (int e = tid; e < m * nb; e += QT) {
        int const i = e % m, j = e / m;
        double const x = P[j * LDP + i];
        X[(size_t)j * ldx + i] = (i <= j) ? x : 0.0;
        V[(size_t)j * ldv + i] = (i > j) ? x * scl[j] : (i == j ? 1.0 : 0.0);
    }
    for (int e = tid; e < nb * nb; e += QT) T[e] = Tl[(e / nb) * R2 + e % nb] * ((e % nb) <= (e / nb) ? 1.0 : 0.0);
}

// Stage 1, right: the mb x m block Mb = B(i1-mb:i1, i0:i1) (mb <= r, m <= 2 r) becomes [0 R] (R mb x mb upper
// triangular) under G = I - V T V^T from the right: QR of the flipped transpose, flip(Mb^T) = Qr R, G = flip Qr flip.
// V: m x mb (rows = columns of the block), T: mb x mb.
__global__ __launch_bounds__(QT) void ht2_rq_kernel(double *__restrict__ Mb, int ldb, int mb, int m,
    double *__restrict__ V, int ldv, double *__restrict__ T)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *P = lds, *Tl = P + R2 * LDP, *G = Tl + R2 * R2, *tau = G + R2 * R2, *scl = tau + R2;
    int const tid = threadIdx.x;
    // P(a, b) = Mb(mb-1-b, m-1-a): m rows, mb columns
    for (int e = tid; e < m * mb; e += QT) {
        int const a = e % m, b = e / m;
        P[b * LDP + a] = Mb[(size_t)(m - 1 - a) * ldb + (mb - 1 - b)];
    }
    __syncthreads();
    wave_qr(P, LDP, m, mb, tau, scl);
    lds_tfactor(P, LDP, m, mb, tau, scl, Tl, R2, G);
    for (int e = tid; e < m * mb; e += QT) {
        int const a = e % m, b = e / m;
        double const x = P[b * LDP + a];
        // Mb_new(q, c) = R(m-1-c, mb-1-q)
        Mb[(size_t)(m - 1 - a) * ldb + (mb - 1 - b)] = (a <= b) ? x : 0.0;
        // V(row = column index of the block) = flipped rows of Vr
        V[(size_t)b * ldv + (m - 1 - a)] = (a > b) ? x * scl[b] : (a == b ? 1.0 : 0.0);
    }
    for (int e = tid; e < mb * mb; e += QT) T[e] = Tl[(e / mb) * R2 + e % mb] * ((e % mb) <= (e / mb) ? 1.0 : 0.0);
}
constexpr int PANEL_LDS_BYTES = (R2 * LDP + 2 * R2 * R2 + 2 * R2 + 16) * 8;

// ---- stage 2 --------------------------------------------------------------------------------------------------
struct Wave2 { int n, tau_idx, jlo, count; };      // wavefront tau_idx: sweeps jlo .. jlo + count - 1, position t = tau_idx - 3 j

__device__ __forceinline__ bool step_of(Wave2 const &w, int k, int &p, int &p1, int &c0)
{
    int const j = w.jlo + k, t = w.tau_idx - 3 * j;
    if (k >= w.count || t < 0 || j > w.n - 3) return false;
    p = j + 1 + t * R2;
    if (p > w.n - 2) return false;
    p1 = min(p + R2, w.n);
    c0 = (t == 0) ? j : p - R2;
    return true;
}

// left reflector of every step of the wavefront: v (HV[k][0:64], v[0] = 1), tau (HT[k]); the column is reduced
__global__ __launch_bounds__(64) void ht2_genh_kernel(Wave2 w, double *__restrict__ A, int lda,
    double *__restrict__ HV, double *__restrict__ HT)
{
    int const k = blockIdx.x, lane = threadIdx.x;
    int p, p1, c0;
    if (!step_of(w, k, p, p1, c0)) { if (lane == 0) HT[k] = 0.0; HV[k * R2 + lane] = 0.0; return; }
    int const len = p1 - p;
    double *col = A + (size_t)c0 * lda + p;
    double const x = lane < len ? col[lane] : 0.0;
    double const ss = wsum64(lane >= 1 ? x * x : 0.0);
    double const alpha = __shfl(x, 0);
    double t = 0.0, beta = alpha, scale = 0.0;
    if (ss != 0.0) { beta = -copysign(sqrt(alpha * alpha + ss), alpha); t = (beta - alpha) / beta; scale = 1.0 / (alpha - beta); }
    HV[k * R2 + lane] = lane == 0 ? 1.0 : (lane < len ? x * scale : 0.0);
    if (lane == 0) HT[k] = t;
    if (lane < len) col[lane] = lane == 0 ? beta : 0.0;
}

// X(p:p1, cb:n) <- (I - tau v v^T) X for X = A (z = 0, cb = c0 + 1) and X = B (z = 1, cb = p): 16 lanes per
// column, four rows each; blockIdx.x: chunk of 256 columns, blockIdx.y: step
__global__ __launch_bounds__(256) void ht2_apply_left_kernel(Wave2 w, double *__restrict__ A, int lda,
    double *__restrict__ B, int ldb, double const *__restrict__ HV, double const *__restrict__ HT)
{
    int const k = blockIdx.y;
    int p, p1, c0;
    if (!step_of(w, k, p, p1, c0)) return;
    double const tau = HT[k];
    if (tau == 0.0) return;
    bool const isB = blockIdx.z == 1;
    double *X = isB ? B : A;
    int const ld = isB ? ldb : lda, cb = isB ? p : c0 + 1, len = p1 - p;
    int const tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;            // 16 column groups per block
    int const r0 = 4 * l16;
    double v[4];
    #pragma unroll
    for (int q = 0; q < 4; q++) v[q] = (r0 + q < len) ? HV[k * R2 + r0 + q] : 0.0;
    int const cbeg = cb + blockIdx.x * 256;
    for (int c = cbeg + grp; c < min(cbeg + 256, w.n); c += 16) {
        double *x = X + (size_t)c * ld + p + r0;
        double y[4], d = 0.0;
        #pragma unroll
        for (int q = 0; q < 4; q++) { y[q] = (r0 + q < len) ? x[q] : 0.0; d += v[q] * y[q]; }
        // sum over the 16 lanes of the column
        d += __shfl_xor(d, 8, 16); d += __shfl_xor(d, 4, 16); d += __shfl_xor(d, 2, 16); d += __shfl_xor(d, 1, 16);
        d *= tau;
        #pragma unroll
        for (int q = 0; q < 4; q++) if (r0 + q < len) x[q] = y[q] - d * v[q];
    }
}

// X(rb:re, p:p1) <- X (I - tau v v^T): one thread per row, the row's entries in registers.
//   side 0 (after the left build): X = Q, all rows.
//   side 1 (after the right build): z = 0: B rows [0, p1) (and the first column of the block cleaned),
//                                   z = 1: A rows [0, min(p1 + r, n)), z = 2: Z all rows.
__global__ __launch_bounds__(256) void ht2_apply_right_kernel(Wave2 w, int side, double *__restrict__ X0, int ld0,
    double *__restrict__ X1, int ld1, double *__restrict__ X2, int ld2, int nrows_q,
    double const *__restrict__ RV, double const *__restrict__ RT)
{
    __shared__ double s_v[R2];
    int const k = blockIdx.y;
    int p, p1, c0;
    if (!step_of(w, k, p, p1, c0)) return;
    double const tau = RT[k];
    int const len = p1 - p, z = blockIdx.z;
    double *X = z == 0 ? X0 : (z == 1 ? X1 : X2);
    int const ld = z == 0 ? ld0 : (z == 1 ? ld1 : ld2);
    if (X == nullptr) return;
    int rows;
    if (side == 0) rows = nrows_q;
    else rows = z == 0 ? p1 : (z == 1 ? min(p1 + R2, w.n) : nrows_q);
    int const row = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= rows) return;
    if (threadIdx.x < R2) s_v[threadIdx.x] = threadIdx.x < len ? RV[k * R2 + threadIdx.x] : 0.0;
    __syncthreads();
    if (row >= rows) return;
    double *x = X + (size_t)p * ld + row;
    if (tau != 0.0) {
        double y[R2], d = 0.0;
        #pragma unroll
        for (int q = 0; q < R2; q++) { y[q] = q < len ? x[(size_t)q * ld] : 0.0; d += y[q] * s_v[q]; }
        d *= tau;
        #pragma unroll
        for (int q = 0; q < R2; q++) if (q < len) x[(size_t)q * ld] = y[q] - d * s_v[q];
    }
    if (side == 1 && z == 0 && row > p && row < p1) x[0] = 0.0;            // B(p+1:p1, p) = 0 exactly
}

// the opposite reflector of every step: x orthogonal to rows 1 .. len-1 of M = B(p:p1, p:p1) (QR of those rows,
// transposed, in LDS; x = the last column of the full Q), then the reflector G = I - tz w w^T with G e_1 = +-x
__global__ __launch_bounds__(QT) void ht2_geng_kernel(Wave2 w, double const *__restrict__ B, int ldb,
    double *__restrict__ GV, double *__restrict__ GT)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int LQ = R2 + 1;
    double *P = lds, *tau = P + R2 * LQ, *scl = tau + R2;
    int const k = blockIdx.x, tid = threadIdx.x;
    int p, p1, c0;
    if (!step_of(w, k, p, p1, c0)) { if (tid == 0) GT[k] = 0.0; if (tid < R2) GV[k * R2 + tid] = 0.0; return; }
    int const len = p1 - p, kq = len - 1;
    // P(a, b) = M(b + 1, a): len rows, len - 1 columns
    for (int idx = tid; idx < len * kq; idx += QT) {
        int const a = idx % len, b = idx / len;
        P[b * LQ + a] = B[(size_t)(p + a) * ldb + p + b + 1];
    }
    __syncthreads();
    wave_qr(P, LQ, len, kq, tau, scl);
    if (tid < 64) {
        // e = H_0 ... H_{kq-1} e_{len-1}, one entry per lane; then the reflector from x = e (dlarfg)
        int const lane = tid;
        double ev = (lane == len - 1) ? 1.0 : 0.0;
        for (int c = kq - 1; c >= 0; c--) {
            double const t = tau[c];
            if (t == 0.0) continue;
            double const vi = (lane > c && lane < len) ? scl[c] * P[c * LQ + lane] : (lane == c ? 1.0 : 0.0);
            double const d = wsum64(vi * ev) * t;
            ev -= d * vi;
        }
        double const x = lane < len ? ev : 0.0;
        double const ss = wsum64(lane >= 1 ? x * x : 0.0);
        double const alpha = __shfl(x, 0);
        double t = 0.0, scale = 0.0;
        if (ss != 0.0) { double const beta = -copysign(sqrt(alpha * alpha + ss), alpha); t = (beta - alpha) / beta; scale = 1.0 / (alpha - beta); }
        GV[k * R2 + lane] = lane == 0 ? 1.0 : (lane < len ? x * scale : 0.0);
        if (lane == 0) GT[k] = t;
    }
}
constexpr int GENG_LDS_BYTES = (R2 * (R2 + 1) + 3 * R2 + 16) * 8;

struct Ht2Workspace {
    int n = 0;
    double *V = nullptr, *T = nullptr, *W1 = nullptr, *W2 = nullptr;
    double *HV = nullptr, *HT = nullptr, *GV = nullptr, *GT = nullptr;
    int maxk = 0;
    bool attr = false;
    void ensure(int n_)
    {
        if (!attr) {
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)ht2_panel_qr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PANEL_LDS_BYTES));
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)ht2_rq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PANEL_LDS_BYTES));
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)ht2_geng_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GENG_LDS_BYTES));
            attr = true;
        }
        if (n_ <= n) return;
        release();
        n = n_;
        maxk = n / (3 * R2 - 1) + 4;
        auto alloc = [](double *&p, size_t count) { SN_HIP_CHECK(hipMalloc((void **)&p, count * sizeof(double))); };
        alloc(V, (size_t)2 * R2 * R2); alloc(T, (size_t)R2 * R2);
        alloc(W1, (size_t)R2 * n); alloc(W2, (size_t)R2 * n);
        alloc(HV, (size_t)maxk * R2); alloc(HT, maxk); alloc(GV, (size_t)maxk * R2); alloc(GT, maxk);
    }
    void release()
    {
        double **all[] = {&V, &T, &W1, &W2, &HV, &HT, &GV, &GT};
        for (double **p : all) if (*p) { SN_HIP_CHECK(hipFree(*p)); *p = nullptr; }
        n = 0;
    }
};
Ht2Workspace g_ht2;

// X (m x ncols) <- (I - V T V^T)^T X
void wy_left(hipStream_t s, Ht2Workspace &ws, int m, int k, int ncols, double *X, int ldx)
{
    if (ncols <= 0 || k <= 0) return;
    dgemm(s, 'T', 'N', k, ncols, m, 1.0, ws.V, 2 * R2, X, ldx, 0.0, ws.W1, R2);
    dgemm(s, 'T', 'N', k, ncols, k, 1.0, ws.T, k, ws.W1, R2, 0.0, ws.W2, R2);
    dgemm(s, 'N', 'N', m, ncols, k, -1.0, ws.V, 2 * R2, ws.W2, R2, 1.0, X, ldx);
}
// X (nrows x m) <- X (I - V T V^T)
void wy_right(hipStream_t s, Ht2Workspace &ws, int nrows, int m, int k, double *X, int ldx)
{
    if (nrows <= 0 || k <= 0) return;
    dgemm(s, 'N', 'N', nrows, k, m, 1.0, X, ldx, ws.V, 2 * R2, 0.0, ws.W1, nrows);
    dgemm(s, 'N', 'N', nrows, k, k, 1.0, ws.W1, nrows, ws.T, k, 0.0, ws.W2, nrows);
    dgemm(s, 'N', 'T', nrows, m, k, -1.0, ws.W2, nrows, ws.V, 2 * R2, 1.0, X, ldx);
}

} // namespace

void ht_two_stage_release_workspace() { g_ht2.release(); }

// (A, B), B upper triangular -> Hessenberg-triangular, Q <- Q U1, Z <- Z U2 (Q, Z may be NULL).  Everything on `s`.
// ms[0], ms[1] (may be NULL): set by the caller from events around the two stages.
int ht_two_stage_device(hipStream_t s, int n, double *A, int lda, double *B, int ldb, double *Q, int ldq,
    double *Z, int ldz, hipEvent_t between)
{
    Ht2Workspace &ws = g_ht2;
    ws.ensure(n);
    int const r = R2;
    // ---- stage 1 -----------------------------------------------------------------------------------------------
    for (int jc = 0; jc < n - r - 1; jc += r) {
        int const nb = std::min(r, n - jc), top = jc + r;
        std::vector<int> starts;
        for (int i = top; i < n; i += r) starts.push_back(i);
        auto left_step = [&](int i0, int i1) {
            int const m = i1 - i0, k = nb;
            hipLaunchKernelGGL(ht2_panel_qr_kernel, dim3(1), dim3(QT), PANEL_LDS_BYTES, s, A + (size_t)jc * lda + i0, lda, m, nb, ws.V, 2 * r, ws.T);
            wy_left(s, ws, m, k, n - jc - nb, A + (size_t)(jc + nb) * lda + i0, lda);
            wy_left(s, ws, m, k, n - i0, B + (size_t)i0 * ldb + i0, ldb);
            if (Q) wy_right(s, ws, n, m, k, Q + (size_t)i0 * ldq, ldq);
        };
        auto right_step = [&](int i0, int i1, int mb) {
            // the bottom mb rows of the block B(i0:i1, i0:i1) become [0 R]
            int const m = i1 - i0;
            hipLaunchKernelGGL(ht2_rq_kernel, dim3(1), dim3(QT), PANEL_LDS_BYTES, s, B + (size_t)i0 * ldb + (i1 - mb), ldb, mb, m, ws.V, 2 * r, ws.T);
            wy_right(s, ws, i1 - mb, m, mb, B + (size_t)i0 * ldb, ldb);
            wy_right(s, ws, n, m, mb, A + (size_t)i0 * lda, lda);
            if (Z) wy_right(s, ws, n, m, mb, Z + (size_t)i0 * ldz, ldz);
        };
        int const K = (int)starts.size();
        for (int k = K - 1; k >= 1; k--) {
            int const i0 = starts[k - 1], i1 = std::min(starts[k] + r, n);
            left_step(i0, i1);
            int const mb = i1 - (i0 + r);
            if (mb > 0) right_step(i0, i1, mb);
        }
        if (K == 1 && n - top > 1) left_step(top, n);
        int const i1 = std::min(top + r, n);
        if (i1 - top > 1) right_step(top, i1, i1 - top);
    }
    if (between) SN_HIP_CHECK(hipEventRecord(between, s));
    // ---- stage 2 -----------------------------------------------------------------------------------------------
    for (int tau_idx = 0;; tau_idx++) {
        int const jhi = std::min(tau_idx / 3, n - 3);
        // position t = tau_idx - 3 j must satisfy j + 1 + t r <= n - 2
        long const num = (long)tau_idx * r - (n - 3);
        int const jlo = num <= 0 ? 0 : (int)((num + (3 * r - 1) - 1) / (3 * r - 1));
        if (jlo > jhi) { if (tau_idx / 3 >= n - 3) break; else continue; }
        int const count = jhi - jlo + 1;
        if (count > ws.maxk) return -1;
        Wave2 const w{n, tau_idx, jlo, count};
        hipLaunchKernelGGL(ht2_genh_kernel, dim3(count), dim3(64), 0, s, w, A, lda, ws.HV, ws.HT);
        hipLaunchKernelGGL(ht2_apply_left_kernel, dim3(divceil(n, 256), count, 2), dim3(256), 0, s, w, A, lda, B, ldb, ws.HV, ws.HT);
        if (Q)
            hipLaunchKernelGGL(ht2_apply_right_kernel, dim3(divceil(n, 256), count, 1), dim3(256), 0, s, w, 0, Q, ldq,
                (double *)nullptr, 0, (double *)nullptr, 0, n, ws.HV, ws.HT);
        hipLaunchKernelGGL(ht2_geng_kernel, dim3(count), dim3(QT), GENG_LDS_BYTES, s, w, B, ldb, ws.GV, ws.GT);
        hipLaunchKernelGGL(ht2_apply_right_kernel, dim3(divceil(n, 256), count, Z ? 3 : 2), dim3(256), 0, s, w, 1, B, ldb,
            A, lda, Z, ldz, n, ws.GV, ws.GT);
    }
    return 0;
}

} // namespace sn
