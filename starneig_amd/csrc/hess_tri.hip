// Hessenberg-triangular reduction of a general pencil (A, B) on one MI355X (row f4 of SURVEY 8f).
//
// Reference: wrappers/lapack.c:45-176 (starneig_GEP_SM_HessenbergTriangular) -- a sequence of LAPACK
// calls: dgeqrf(B), dormqr on A and Q, B <- R, dgghd3.  Kept: the two steps and their results
// (B = Q0 R by Householder reflectors with the dlarfg conventions; then the rotation-based
// reduction of Moler & Stewart: column by column, A(i, j) is annihilated from the bottom by a
// rotation of rows (i-1, i) and the fill-in B(i, i-1) by a rotation of columns (i-1, i), dlartg
// conventions, so that H, T, Q, Z agree with LAPACK's dgghrd up to rounding).
// Re-designed for the GPU:
//  * QR step: panels of 64 columns; one launch per column finishes the previous reflector, applies
//    it to the rest of the panel and accumulates the sums the next reflector needs (its norm and its
//    inner products with the remaining columns, taken on the unscaled column so that they do not
//    wait for the norm); per-workgroup partial sums, reduced in a fixed order.  Trailing matrix,
//    A and Q are updated with the compact-WY form on the fp64 MFMA GEMM.
//  * Rotation step, per column j ("sweep"):
//      scan     all row rotations of the sweep at once: the eliminated entries are the suffix sums
//               of squares of the column (a parallel scan instead of n dependent dlartg calls);
//      row pass the row rotations on A and B (one lane per column, LDS-transposed tiles);
//      chain    the fill-in of B (now upper Hessenberg) is removed by the column rotations, a
//               dependent chain along the diagonal: one workgroup per group of 256 rows, one lane
//               per row, the diagonal tile in LDS, the rotation built from two lane reads; the
//               other waves of the workgroup follow block by block;
//      col pass the column rotations of a finished group on the rows above (B), on A and on Z,
//               one lane per row, coalesced column loads; Q takes the row rotations the same way.
//    The chain is what bounds the step (n^2/2 dependent rotations); everything else streams.
#include "common.h"
#include "tuning.h"
#include <starneig/error.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <vector>

namespace sn {

namespace {

// ---------------------------------------------------------------------------------------------
// QR step
// ---------------------------------------------------------------------------------------------
constexpr int QNB = 64;             // panel width
constexpr int QROWS = 256;          // rows of one workgroup of the panel kernel
constexpr int QP = QNB;             // partial record of one workgroup: index k = dot with column k, index jj = sum of squares

// sum over the 16 lanes of a DPP row (every lane of the row gets it): VALU only, no LDS crossbar and no wait -- the
// 64 sums of one launch of the panel kernel interleave (a chain of six __shfl_xor steps each cost 0.55 us a column)
template <int CTRL>
__device__ __forceinline__ double qr_dpp(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double x)
{
    x += qr_dpp<0x128>(x);   // row_ror:8
    x += qr_dpp<0x124>(x);   // row_ror:4
    x += qr_dpp<0x122>(x);   // row_ror:2
    x += qr_dpp<0x121>(x);   // row_ror:1
    return x;
}

// Launch jj = 0..nb of one panel (rows/columns from p0):
//   jj > 0:  reflector jj-1 is finished from the sums of launch jj-1 (dlarfg: beta, tau, v = x / (alpha - beta))
//            and applied to columns jj..nb-1;
//   jj < nb: sums of column jj on the updated data: S = sum x_i^2 and D_k = sum x_i a_ik (rows below
//            the diagonal), and the pivot row itself (prow), so that no workgroup reads what another
//            one writes in the same launch.
// A workgroup is QROWS rows x QCG column groups (1024 threads): the thread of row r and group cg holds the columns
// k = QCG u + cg of its row in registers -- every load of a launch in flight at once, QNB / QCG dependent steps a thread
// instead of QNB.  (Round 6.  Before: one thread a row, a loop of load - update - store - sum over the columns: the stores
// to B kept the next load behind them and each of the 64 sums was a chain of six __shfl_xor steps -- 0.55 us per
// remaining column, 24.5 us a launch at n = 8000.)
constexpr int QCG = 4, QCPT = QNB / QCG;
__global__ __launch_bounds__(QROWS * QCG) void ht_qr_col_kernel(int n, int p0, int nb, int jj,
    double *__restrict__ B, int ldb, double *__restrict__ Vp, int ldp, double *__restrict__ tau,
    double const *__restrict__ part_in, double *__restrict__ part_out,
    double const *__restrict__ prow_in, double *__restrict__ prow_out, int nwg)
{
    __shared__ double w[QNB];
    __shared__ double red[QNB][QROWS / 16 + 1];        // one partial per DPP row of 16 lanes
    __shared__ double xs[QROWS];                        // column jj of the workgroup's rows, by the group that owns it
    __shared__ double sumsq;
    int const tid = threadIdx.x, row = tid & (QROWS - 1), cg = tid / QROWS;
    int const i = p0 + blockIdx.x * QROWS + row;        // global row
    bool const valid = i < n;
    double coef = 0.0;
    int kind = 0;                                       // 1: pivot row, 2: below it
    if (jj > 0) {
        int const pc = jj - 1, pd = p0 + pc;
        double acc = 0.0;
        if (tid >= pc && tid < nb) {
            int g = 0;
            for (; g + 8 <= nwg; g += 8) {              // (eight loads in flight, added in workgroup order)
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = part_in[(size_t)(g + u) * QP + tid];
#pragma unroll
                for (int u = 0; u < 8; u++) acc += v[u];
            }
            for (; g < nwg; g++) acc += part_in[(size_t)g * QP + tid];
        }
        if (tid == pc) sumsq = acc;
        size_t const off = (size_t)(p0 + pc) * ldb + i;
        double const braw = (valid && i > pd) ? B[off] : 0.0;     // read by all four groups of the row before group 0 rewrites it
        __syncthreads();
        double const S = sumsq, alpha = prow_in[pc];
        double beta = alpha, t = 0.0, scale = 0.0;
        if (S != 0.0) {
            beta = -copysign(sqrt(alpha * alpha + S), alpha);
            t = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        if (tid >= jj && tid < nb) w[tid] = t * (prow_in[tid] + scale * acc);
        __syncthreads();
        if (valid) {
            double v = 0.0;
            if (i > pd) { coef = braw * scale; kind = 2; v = coef; }
            else if (i == pd) { kind = 1; v = 1.0; }
            if (cg == 0) {
                if (kind == 2) B[off] = 0.0;
                else if (kind == 1) { B[off] = beta; tau[pc] = t; }
                Vp[(size_t)pc * ldp + (i - p0)] = v;
            }
        }
    }
    if (jj < nb) {
        // (No divergent branch assigns to the register array: loads from clamped addresses and selects -- with
        // conditional assignments the compiler kept the array as one 32-register value and spilled it at every merge.)
        int const pdn = p0 + jj;
        double a[QCPT];
        bool act[QCPT];
        double const *Bi = B + (size_t)p0 * ldb + min(i, n - 1);
        double const c = kind == 2 ? coef : (kind == 1 ? 1.0 : 0.0);
#pragma unroll
        for (int u = 0; u < QCPT; u++) {
            int const k = QCG * u + cg;
            act[u] = valid && k >= jj && k < nb;
            double const v = Bi[(size_t)min(max(k, jj), nb - 1) * ldb];
            a[u] = act[u] ? v : 0.0;
        }
#pragma unroll
        for (int u = 0; u < QCPT; u++) {
            int const k = QCG * u + cg;
            double const wk = w[k];
            a[u] -= c * ((jj > 0 && act[u]) ? wk : 0.0);
            if (act[u] && kind) B[(size_t)(p0 + k) * ldb + i] = a[u];
        }
        if (cg == (jj & (QCG - 1))) {
            double x = 0.0;
#pragma unroll
            for (int u = 0; u < QCPT; u++) x = (u == jj / QCG) ? a[u] : x;
            xs[row] = (valid && i > pdn) ? x : 0.0;
        }
        if (valid && i == pdn) {
#pragma unroll
            for (int u = 0; u < QCPT; u++) { int const k = QCG * u + cg; if (act[u]) prow_out[k] = a[u]; }
        }
        __syncthreads();
        double const xn = xs[row];
#pragma unroll
        for (int u = 0; u < QCPT; u++) {
            int const k = QCG * u + cg;
            if (k >= jj && k < nb) {
                double const p = row16_sum(xn * a[u]);
                if ((row & 15) == 0) red[k][row >> 4] = p;
            }
        }
        __syncthreads();
        if (tid >= jj && tid < nb) {
            double p = 0.0;
            for (int q = 0; q < QROWS / 16; q++) p += red[tid][q];
            part_out[(size_t)blockIdx.x * QP + tid] = p;
        }
    }
}

// T of the compact-WY form (dlarft, forward / columnwise) from G = V^T V and tau
__global__ __launch_bounds__(64) void ht_qr_tfactor_kernel(int nb, double const *__restrict__ G,
    double const *__restrict__ tau, double *__restrict__ T)
{
    __shared__ double t[QNB][QNB + 1], g[QNB][QNB + 1];
    int const i = threadIdx.x;
    for (int k = 0; k < QNB; k++) {
        g[i][k] = (i < nb && k < nb) ? G[(size_t)k * QNB + i] : 0.0;
        t[i][k] = 0.0;
    }
    __syncthreads();
    for (int k = 0; k < nb; k++) {
        double const tk = tau[k];
        double acc = 0.0;
        for (int l = i; l < k; l++) acc += t[i][l] * g[l][k];
        t[i][k] = (i < k) ? -tk * acc : (i == k ? tk : 0.0);
    }
    for (int k = 0; k < QNB; k++) T[(size_t)k * QNB + i] = t[i][k];
}

// ---------------------------------------------------------------------------------------------
// Rotation step
// ---------------------------------------------------------------------------------------------
constexpr int HG = 8;              // waves of the chain workgroup that hold the rows of the group
constexpr int HF = 0;               // follower waves: the 64*HF rows above the group
constexpr int HGR = 64 * HG;        // rows of one diagonal group
constexpr int HGR_MAX = 512;
static_assert(HGR <= HGR_MAX, "the LDS column pass holds one group");

typedef double v2d __attribute__((ext_vector_type(2)));

__device__ inline double readlane_d(double v, int l)
{
    int const lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int const hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// LAPACK >= 3.10 dlartg conventions: [c s; -s c] (f, g)^T = (r, 0)^T, c >= 0, r = c f + s g carries the
// sign of f.  On the chain every instruction counts (a wave64 VALU instruction occupies the SIMD
// for 4 cycles, a dependent fp64 result takes 6) and a branch on a fresh VALU result costs as much as
// ten of them (measured: 66 ns for LAPACK's safe-range test): 1/sqrt(f^2+g^2) from v_rsq_f64
// (2^-24) and ONE third-order Newton step (measured residual 1.6e-16 over the range), no division,
// no sqrt expansion, one select.  Range: B is scaled to max|b| in [1, 2) on entry
// (hessenberg_triangular_device), rotations preserve the Frobenius norm, so f^2 + g^2 cannot
// overflow; where it underflows (both below 1e-140 of the largest entry) the fill-in is dropped, a
// backward error far below u.
__device__ inline void ht_lartg(double f, double g, double &c, double &s)
{
    double const h2 = fma(f, f, g * g);
    double y = __builtin_amdgcn_rsq(h2);
    double const e1 = fma(-h2 * y, y, 1.0);
    y = fma(y * e1, fma(0.375, e1, 0.5), y);            // y (1 + e/2 + 3 e^2 / 8)
    bool const trivial = h2 < 1e-280;                   // also f = g = 0 (y = inf)
    c = trivial ? 1.0 : fabs(f) * y;
    s = trivial ? 0.0 : g * copysign(y, f);
}

// All row rotations of sweep j from column j of A: rotation i (rows i-1, i), i = n-1 .. j+2, is
// dlartg(a_{i-1}, r_i) with r_i = +-sqrt(sum_{p >= i} a_p^2) -- a suffix scan.  The column is
// overwritten with (r_{j+1}, 0, ..., 0).
__global__ __launch_bounds__(1024) void ht_scan_kernel(int n, int j, double *__restrict__ A, int lda,
    double *__restrict__ Rc, double *__restrict__ Rs)
{
    __shared__ double sh[1024];
    int const tid = threadIdx.x;
    int const m = n - j - 1;
    double *a = A + (size_t)j * lda + (j + 1);
    int const per = (m + 1023) / 1024;
    int const q0 = std::min(m, tid * per), q1 = std::min(m, q0 + per);
    double mx = 0.0;
    for (int q = q0; q < q1; q++) mx = fmax(mx, fabs(a[q]));
    sh[tid] = mx;
    __syncthreads();
    for (int off = 512; off; off >>= 1) {
        if (tid < off) sh[tid] = fmax(sh[tid], sh[tid + off]);
        __syncthreads();
    }
    double const amax = sh[0];
    __syncthreads();
    if (amax == 0.0) {
        for (int q = std::max(q0, 1); q < q1; q++) { Rc[j + 1 + q] = 1.0; Rs[j + 1 + q] = 0.0; }
        return;
    }
    int const e = ilogb(amax);
    double tot = 0.0;
    for (int q = q1 - 1; q >= q0; q--) { double const v = scalbn(a[q], -e); tot += v * v; }
    sh[tid] = tot;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          // inclusive suffix scan of the chunk totals
        double const v = (tid + off < 1024) ? sh[tid + off] : 0.0;
        __syncthreads();
        sh[tid] += v;
        __syncthreads();
    }
    double Sq = (tid + 1 < 1024) ? sh[tid + 1] : 0.0;   // sum over the chunks behind this one
    double r0 = 0.0;
    for (int q = q1 - 1; q >= q0; q--) {
        double const aq = scalbn(a[q], -e);
        Sq += aq * aq;                                  // S[q] = sum_{p >= q} a_p^2
        double const g = (q == m - 1) ? aq : (aq < 0.0 ? -sqrt(Sq) : sqrt(Sq));   // r_q
        if (q >= 1) {
            double const f = scalbn(a[q - 1], -e);
            double c, s;
            if (g == 0.0) { c = 1.0; s = 0.0; }
            else if (f == 0.0) { c = 0.0; s = copysign(1.0, g); }
            else { double const d = sqrt(f * f + Sq); c = fabs(f) / d; s = g / copysign(d, f); }
            Rc[j + 1 + q] = c;
            Rs[j + 1 + q] = s;
        } else
            r0 = scalbn(g, e);
    }
    __syncthreads();
    for (int q = q0; q < q1; q++) a[q] = (q == 0) ? r0 : 0.0;
}

// Row rotations i = n-1 .. j+2 (rows i-1, i) on the columns of M, segment-parallel.  Along a column
// the rotations are a recurrence on one carried value (the current entry of the lower row):
//     out(i) = c_i carry - s_i up,   carry <- c_i up + s_i carry,   up = M(i-1, col).
// Over a segment of 64 rows it is affine in the incoming carry: carry_out = alpha carry_in + beta,
// alpha = prod s_i (the same for every column), beta = the recurrence started from 0.  Three passes,
// every 64 x 64 tile an independent workgroup in the first and the last:
//   pass 1  beta of every tile, and a copy of the tile's first row (the tile above overwrites it);
//   pass 2  per column, the carries entering each segment (a scan over <= n/64 segments);
//   pass 3  the recurrence again with the right incoming carry, results written in place.
// tri: M is upper triangular (B): column c starts at rotation c+1, which creates the fill-in M(c+1, c);
// tiles below the diagonal are skipped.  Segment s holds the "upper" rows 64 s .. 64 s + 63.
struct RowPassArgs {
    double *M; int ld; int n, j, c_begin, tri;
    double const *Rc, *Rs;
    double *beta, *uprow, *alpha;       // [segment][column], [segment][column], [segment]
    int seg_lo;                         // first segment with rotations (contains row j+1)
};

__device__ inline bool rowpass_tile(RowPassArgs const &a, int strip, int seg, int &c0, int &rlo, int &rhi)
{
    c0 = a.c_begin + 64 * strip;
    rlo = std::max(64 * seg, a.j + 1);
    rhi = std::min(64 * seg + 63, a.n - 2);
    if (rhi < rlo) return false;
    if (a.tri && rlo > std::min(a.n - 1, c0 + 63)) return false;     // below the diagonal: nothing to rotate
    return true;
}

// the recurrence over one tile; lane = column.  WRITE: results into the LDS tile (row k holds out(rlo+k+1))
template <bool WRITE>
__device__ inline double rowpass_recur(RowPassArgs const &a, double (*tile)[65], double const *lc, double const *ls,
    int lane, int c, int rlo, int cnt, double carry, double up0)
{
    int const istart = a.tri ? std::min(a.n - 1, c + 1) : a.n - 1;
#pragma unroll 8
    for (int k = cnt - 1; k >= 0; k--) {
        double const up = (k == 0) ? up0 : tile[lane][k], cr = lc[k], sr = ls[k];
        bool const on = rlo + k + 1 <= istart;
        double const out = cr * carry - sr * up, nc = cr * up + sr * carry;
        if (WRITE) tile[lane][k] = on ? out : 0.0;
        carry = on ? nc : carry;
    }
    return carry;
}

__global__ __launch_bounds__(64) void ht_rowpass1_kernel(RowPassArgs a)
{
    __shared__ double tile[64][65];
    __shared__ double lc[64], ls[64];
    int const lane = threadIdx.x, seg = a.seg_lo + blockIdx.y;
    int c0, rlo, rhi;
    if (!rowpass_tile(a, blockIdx.x, seg, c0, rlo, rhi)) return;
    int const cnt = rhi - rlo + 1, ncol = std::min(64, a.n - c0), c = c0 + lane;
    double const *src = a.M + (size_t)c0 * a.ld + rlo + lane;
    double v[64];
#pragma unroll
    for (int cc = 0; cc < 64; cc++) v[cc] = (lane < cnt && cc < ncol) ? src[(size_t)cc * a.ld] : 0.0;
    if (lane < cnt) { lc[lane] = a.Rc[rlo + 1 + lane]; ls[lane] = a.Rs[rlo + 1 + lane]; }
#pragma unroll
    for (int cc = 0; cc < 64; cc++) tile[cc][lane] = v[cc];
    __syncthreads();
    if (c < a.n) {
        double const up0 = tile[lane][0];
        a.uprow[(size_t)seg * a.n + c] = up0;
        a.beta[(size_t)seg * a.n + c] = rowpass_recur<false>(a, tile, lc, ls, lane, c, rlo, cnt, 0.0, up0);
    }
    if (blockIdx.x == gridDim.x - 1 && lane == 0) {     // (the last strip is never below the diagonal)
        double p = 1.0;
        for (int k = 0; k < cnt; k++) p *= ls[k];
        a.alpha[seg] = p;
    }
}

// per column: beta[seg][c] <- carry entering segment seg (from the bottom segment upwards)
__global__ __launch_bounds__(64) void ht_rowpass2_kernel(RowPassArgs a)
{
    int const c = a.c_begin + blockIdx.x * 64 + threadIdx.x;
    if (c >= a.n) return;
    int const istart = a.tri ? std::min(a.n - 1, c + 1) : a.n - 1;       // first rotation of this column
    double carry = (a.tri && istart == c + 1) ? 0.0 : a.M[(size_t)c * a.ld + istart];
    int seg = (istart - 1) / 64;
    for (; seg - 15 >= a.seg_lo; seg -= 16) {               // loads of a batch first (they do not depend on the carry)
        double b[16], al[16];
#pragma unroll
        for (int k = 0; k < 16; k++) { b[k] = a.beta[(size_t)(seg - k) * a.n + c]; al[k] = a.alpha[seg - k]; }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            a.beta[(size_t)(seg - k) * a.n + c] = carry;
            carry = al[k] * carry + b[k];
        }
    }
    for (; seg >= a.seg_lo; seg--) {
        double const b = a.beta[(size_t)seg * a.n + c];
        a.beta[(size_t)seg * a.n + c] = carry;
        carry = a.alpha[seg] * carry + b;
    }
}

__global__ __launch_bounds__(64) void ht_rowpass3_kernel(RowPassArgs a)
{
    __shared__ double tile[64][65];
    __shared__ double lc[64], ls[64];
    int const lane = threadIdx.x, seg = a.seg_lo + blockIdx.y;
    int c0, rlo, rhi;
    if (!rowpass_tile(a, blockIdx.x, seg, c0, rlo, rhi)) return;
    int const cnt = rhi - rlo + 1, ncol = std::min(64, a.n - c0), c = c0 + lane;
    {
        double const *src = a.M + (size_t)c0 * a.ld + rlo + lane;
        double v[64];
#pragma unroll
        for (int cc = 0; cc < 64; cc++) v[cc] = (lane < cnt && lane > 0 && cc < ncol) ? src[(size_t)cc * a.ld] : 0.0;
        if (lane < cnt) { lc[lane] = a.Rc[rlo + 1 + lane]; ls[lane] = a.Rs[rlo + 1 + lane]; }
#pragma unroll
        for (int cc = 0; cc < 64; cc++) tile[cc][lane] = v[cc];
    }
    __syncthreads();
    double carry = 0.0;
    if (c < a.n)
        carry = rowpass_recur<true>(a, tile, lc, ls, lane, c, rlo, cnt, a.beta[(size_t)seg * a.n + c], a.uprow[(size_t)seg * a.n + c]);
    __syncthreads();
    {
        double *dst = a.M + (size_t)c0 * a.ld + rlo + 1 + lane;
        if (lane < cnt)
#pragma unroll 16
            for (int cc = 0; cc < ncol; cc++) dst[(size_t)cc * a.ld] = tile[cc][lane];
    }
    if (rlo == a.j + 1 && c < a.n) a.M[(size_t)c * a.ld + a.j + 1] = carry;
}

// Column rotations t = t_hi .. t_lo on one row per lane: x = M(row, t-1), y = M(row, t) (carried):
//   M(row, t) <- sgn*s*x + c*y,  carry <- c*x - sgn*s*y.
// sgn = +1: the column rotations of B (applied to B, A, Z); sgn = -1: the row rotations, on Q.
template <typename CS>
__device__ inline double ht_apply_cols(double *__restrict__ M, int ld, int row, int t_hi, int t_lo,
    CS cs, double sgn, double y)
{
    int t = t_hi;
    for (; t - 31 >= t_lo; t -= 32) {
        double x[32];
#pragma unroll
        for (int k = 0; k < 32; k++) x[k] = M[(size_t)(t - k - 1) * ld + row];
#pragma unroll
        for (int k = 0; k < 32; k++) {
            double const c = cs.c(t - k), s = sgn * cs.s(t - k);
            M[(size_t)(t - k) * ld + row] = s * x[k] + c * y;
            y = c * x[k] - s * y;
        }
    }
    for (; t >= t_lo; t--) {
        double const x = M[(size_t)(t - 1) * ld + row];
        double const c = cs.c(t), s = sgn * cs.s(t);
        M[(size_t)t * ld + row] = s * x + c * y;
        y = c * x - s * y;
    }
    return y;
}

struct CsGlobal {
    double const *__restrict__ cp, *__restrict__ sp;
    __device__ double c(int t) const { return cp[t]; }
    __device__ double s(int t) const { return sp[t]; }
};

struct ColJob { double *M; int ld; int rows; };

__global__ __launch_bounds__(64) void ht_colpass_kernel(ColJob j0, ColJob j1, ColJob j2, int t_hi, int t_lo,
    double const *__restrict__ Cc, double const *__restrict__ Cs, double sgn)
{
    ColJob const job = blockIdx.y == 0 ? j0 : (blockIdx.y == 1 ? j1 : j2);
    int const row = blockIdx.x * 64 + threadIdx.x;
    if (row >= job.rows || t_hi < t_lo) return;
    double y = job.M[(size_t)t_hi * job.ld + row];
    y = ht_apply_cols(job.M, job.ld, row, t_hi, t_lo, CsGlobal{Cc, Cs}, sgn, y);
    job.M[(size_t)(t_lo - 1) * job.ld + row] = y;
}

// The rotations of TWO consecutive sweeps (a = j-1, b = j) on Q (row rotations) or Z (column rotations) in
// one pass over the columns: Q and Z are pure HBM traffic here, nothing waits for them, and rotation B_k
// only needs A_k and A_{k-1} done.
// Per step k = n-1 .. j+1: A_k on columns (k-1, k), then B_{k+1} on (k, k+1), whose column k+1 is final.
__global__ __launch_bounds__(64) void ht_qpass2_kernel(double *__restrict__ Q, int ld, int rows, int n, int j,
    double const *__restrict__ Rca, double const *__restrict__ Rsa,
    double const *__restrict__ Rcb, double const *__restrict__ Rsb, double sgn)
{
    int const row = blockIdx.x * 64 + threadIdx.x;
    if (row >= rows) return;
    double p = Q[(size_t)(n - 1) * ld + row], q = 0.0;              // column k carried by sweep a, column k+1 by sweep b
    int k = n - 1;
    auto step = [&](int kk, double x) {
        // A_kk on (x = column kk-1, p = column kk); Q takes the row rotations transposed (sgn = -1),
        // Z the column rotations as they are (sgn = +1)
        double const ca = Rca[kk], sa = sgn * Rsa[kk];
        double const colk = sa * x + ca * p;                        // column kk, final for sweep a
        p = ca * x - sa * p;
        if (kk + 1 <= n - 1) {                                      // B_{kk+1} on (column kk, column kk+1)
            double const cb = Rcb[kk + 1], sb = sgn * Rsb[kk + 1];
            Q[(size_t)(kk + 1) * ld + row] = sb * colk + cb * q;
            q = cb * colk - sb * q;
        } else
            q = colk;
    };
    for (; k - 15 >= j + 1; k -= 16) {
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = Q[(size_t)(k - i - 1) * ld + row];
#pragma unroll
        for (int i = 0; i < 16; i++) step(k - i, x[i]);
    }
    for (; k >= j + 1; k--) step(k, Q[(size_t)(k - 1) * ld + row]);
    Q[(size_t)(j + 1) * ld + row] = q;                              // column j+1 carried by sweep b
    Q[(size_t)j * ld + row] = p;                                    // column j carried by sweep a
}

// The same column rotations for the rows of B that the NEXT chain launch waits for: the plain kernel
// above spends one memory latency per batch of columns on every row (50 us for 512 rotations however
// few rows there are).  Here a workgroup stages 32 rows x all columns of the group in LDS with every
// load in flight at once, one half-wave runs the 512-step recurrence out of LDS, and the block goes
// back in one sweep of stores.
constexpr int CL_ROWS = 32, CL_COLS = HGR_MAX + 1;
constexpr int CL_NPH = 256 / CL_ROWS;                   // column phases of the loads = segments of the recurrence
constexpr int CL_LDS_BYTES = (CL_COLS * CL_ROWS + 2 * HGR_MAX + 2 * CL_NPH * CL_ROWS) * 8;
// The recurrence along a row -- rotation i reads the ORIGINAL column i and the carried value y, writes
// column i+1 and passes y on -- is affine in the carry over any run of rotations (y_out = a y_in + b, the
// trick of the row pass).  So it does not have to be ONE chain of up to 512 dependent steps on a
// half-wave while seven others wait: the rotations are cut into CL_NPH segments, every half-wave runs its
// segment once from a zero carry to get (a, b), the incoming carries follow from at most seven
// multiply-adds, and every half-wave runs its segment again with the right carry: 2 x 64 + 7 dependent
// steps instead of 512.  (Each segment's results are those of the serial recurrence from its incoming
// carry; the carries differ from the serial ones by the rounding of a y + b.)
__global__ __launch_bounds__(256) void ht_colpass_lds_kernel(double *__restrict__ M, int ld, int rows,
    int t_hi, int t_lo, double const *__restrict__ Cc, double const *__restrict__ Cs)
{
    extern __shared__ double lds[];
    double (*tile)[CL_ROWS] = (double (*)[CL_ROWS])lds;             // [column t - (t_lo - 1)][row]
    v2d *cs = (v2d *)(lds + CL_COLS * CL_ROWS);                     // [t - t_lo]
    v2d (*seg)[CL_ROWS] = (v2d (*)[CL_ROWS])(lds + CL_COLS * CL_ROWS + 2 * HGR_MAX);    // (a, b) of a segment, per row
    int const tid = threadIdx.x, r = tid & (CL_ROWS - 1), sub = tid / CL_ROWS;      // NPH column phases
    constexpr int NPH = CL_NPH, NLD = (CL_COLS + 2 * NPH - 1) / (2 * NPH);          // two batches of NLD loads
    int const row = blockIdx.x * CL_ROWS + r;
    int const ncol = t_hi - t_lo + 2;                               // columns t_lo - 1 .. t_hi
    bool const rv = row < rows;
    double const *src = M + (size_t)(t_lo - 1) * ld + row;
    for (int c0 = sub; c0 < ncol; c0 += NPH * NLD) {
        double v[NLD];
#pragma unroll
        for (int i = 0; i < NLD; i++) v[i] = (rv && c0 + NPH * i < ncol) ? src[(size_t)(c0 + NPH * i) * ld] : 0.0;
#pragma unroll
        for (int i = 0; i < NLD; i++) if (c0 + NPH * i < ncol) tile[c0 + NPH * i][r] = v[i];
    }
    for (int i = tid; i < ncol - 1; i += 256) cs[i] = v2d{Cc[t_lo + i], Cs[t_lo + i]};
    __syncthreads();
    // rotation t = t_lo + i acts on columns i (t-1) and i+1 (t); segment `sub` holds i = hi .. lo, right to left
    int const nrot = ncol - 1, seglen = (nrot + NPH - 1) / NPH;
    int const hi = nrot - 1 - sub * seglen, lo = max(hi - seglen + 1, 0);
    {
        double a = 1.0, b = 0.0;
        int i = hi;
        for (; i - 7 >= lo; i -= 8) {                               // reads of a batch first
            double x[8]; v2d k[8];
#pragma unroll
            for (int q = 0; q < 8; q++) { x[q] = tile[i - q][r]; k[q] = cs[i - q]; }
#pragma unroll
            for (int q = 0; q < 8; q++) { b = k[q].x * x[q] - k[q].y * b; a = -k[q].y * a; }
        }
        for (; i >= lo; i--) { double const x = tile[i][r]; v2d const k = cs[i]; b = k.x * x - k.y * b; a = -k.y * a; }
        seg[sub][r] = v2d{a, b};
    }
    // the originals that another half-wave overwrites: column `lo` (the left neighbour's first store) and the
    // incoming carry of the whole recurrence (the first store of segment 0)
    double const xlo = (hi >= lo) ? tile[lo][r] : 0.0;
    double y = tile[ncol - 1][r];
    __syncthreads();
    for (int q = 0; q < sub; q++) { v2d const m = seg[q][r]; y = m.x * y + m.y; }
    if (hi >= lo) {
        int i = hi;
        for (; i - 7 > lo; i -= 8) {
            double x[8]; v2d k[8];
#pragma unroll
            for (int q = 0; q < 8; q++) { x[q] = tile[i - q][r]; k[q] = cs[i - q]; }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                tile[i - q + 1][r] = k[q].y * x[q] + k[q].x * y;
                y = k[q].x * x[q] - k[q].y * y;
            }
        }
        for (; i >= lo; i--) {
            double const x = (i == lo) ? xlo : tile[i][r];
            v2d const k = cs[i];
            tile[i + 1][r] = k.y * x + k.x * y;
            y = k.x * x - k.y * y;
        }
        if (lo == 0) tile[0][r] = y;
    }
    __syncthreads();
    if (rv) {
        double *dst = M + (size_t)(t_lo - 1) * ld + row;
        for (int c = sub; c < ncol; c += NPH) dst[(size_t)c * ld] = tile[c][r];
    }
}

// The chain of one diagonal group, rows [g0, g1): B is upper Hessenberg there (fill-in of the row
// pass); rotation t = g1-1 .. max(g0, j+2) of columns (t-1, t) is dlartg(B(t,t), B(t,t-1)) on the
// CURRENT entries.  One lane per row; a lane carries the current value of column t of its row.
// The wave that holds row t builds the rotation (two lane reads) and applies it to its rows; when
// it has finished its 64 rotations the waves of the rows above apply them, then the chain moves
// into the next wave.  Rows above the group: ht_colpass_kernel.
constexpr int CH = 16;              // rotations per publication / per follower step
constexpr int NTB = 3;              // diagonal tiles resident in LDS
constexpr int CHAIN_LDS_DOUBLES(int) { return NTB * 65 * 64; }

// The control words of the chain workgroup live in LDS and are accessed with explicit ds
// instructions: a C++ volatile or atomic access through a generic pointer becomes a FLAT
// instruction with system scope followed by s_waitcnt vmcnt(0) -- a wait for every load and store
// of the wave, on the chain.
__device__ inline unsigned lds_addr(void const *p) { return (unsigned)(size_t)p; }
__device__ inline void lds_store(unsigned addr, int v)
{
    asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory");
}
__device__ inline int lds_load(unsigned addr)
{
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
// Polling wait on control word `which`.  Bounded: a logic error must not hang the GPU -- after
// ~0.2 s the wait gives up and raises the abort flag (word 3), which ends every other wait at once
// (the results are then wrong and the residual checks of the callers / tests say so).
__device__ inline void ht_wait(unsigned ctr, int which, int need)
{
    int spins = 0;
    while (lds_load(ctr + 4 * which) < need && lds_load(ctr + 12) == 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 21)) lds_store(ctr + 12, 1);
    }
}

// One workgroup, three kinds of waves, no barrier after the start and no fence (both would make a
// wave wait for all of its loads and stores in flight):
//  * G "group" waves hold the rows of the group (one lane per row, the current value of column t of
//    the row in a register).  The wave that holds row t is the chain wave: it reads column t-1 of its
//    64 x 64 diagonal tile from LDS, builds the rotation from two lane reads, writes the result back
//    to LDS and the rotation to an LDS table -- it issues no global memory instruction at all.  Every
//    CH rotations it bumps a counter.  Until its turn comes, a group wave is a follower.
//  * F "follower" waves hold the 64*F rows above the group (group waves do the same for the blocks
//    above their own): they poll the counter and apply the published rotations to their rows, CH at
//    a time, with the columns they need already in registers (loaded one step ahead).
//  * one "loader" wave brings the diagonal tiles from HBM to LDS (two blocks ahead of the chain) and
//    writes the finished ones, and the rotations, back.
template <int G, int F, int DBG = 0>
__global__ __launch_bounds__(64 * (G + F + 1)) void ht_chain_kernel(int n, int j, int g0, int g1,
    double *__restrict__ B, int ldb, double *__restrict__ Cc, double *__restrict__ Cs, long long *ts = nullptr,
    int *err = nullptr)
{
    extern __shared__ double lds[];
    double (*tile)[65][64] = (double (*)[65][64])lds;                   // [buffer][column slot][row]
    __shared__ double rot[64 * G][2];                                   // (c, s) of rotation t at t_hi - t
    __shared__ int ctr_words[4];                                        // 0: rotations published, 1: tiles loaded, 2: blocks finished, 3: abort
    unsigned const ctr = lds_addr(ctr_words);
    int const wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int const t_hi = g1 - 1, t_lo = std::max(g0, j + 2);
    int const wtop = std::min(G - 1, (t_hi - g0) / 64), wbot = (t_lo - g0) / 64;
    int const K = wtop - wbot + 1;                                      // blocks with rotations, k = wtop - wb
    if (threadIdx.x == 0) { ctr_words[0] = 0; ctr_words[1] = 1; ctr_words[2] = 0; ctr_words[3] = 0; }
    {
        // the first diagonal tile: all waves of the workgroup fetch it together (the chain cannot start
        // before it is in LDS; the loader alone needs four rounds of loads for it)
        int const b0 = g0 + 64 * wtop, q_hi = std::min(63, t_hi - b0), q_lo = std::max(0, t_lo - b0);
        int const nw = G + F + 1, per = (q_hi - q_lo + nw) / nw;          // columns per wave
        double const *src = B + (size_t)(b0 - 1) * ldb + b0 + lane;
        for (int off = 0; off < per; off += 16) {
            int const c0 = q_lo + per * wv + off;
            double v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = (off + i < per && c0 + i <= q_hi && lane <= q_hi) ? src[(size_t)(c0 + i) * ldb] : 0.0;
#pragma unroll
            for (int i = 0; i < 16; i++) if (off + i < per && c0 + i <= q_hi) tile[0][c0 + i][lane] = v[i];
        }
    }
    __syncthreads();
    if ((DBG & 32) && ts && threadIdx.x == 0) ts[63] = wall_clock64();

    if (wv == G + F) {
        // ---- loader ----
        auto range = [&](int k, int &b0, int &q_lo, int &q_hi) {
            b0 = g0 + 64 * (wtop - k);
            q_hi = std::min(63, t_hi - b0); q_lo = std::max(0, t_lo - b0);
        };
        auto load = [&](int k) {
            int b0, q_lo, q_hi; range(k, b0, q_lo, q_hi);
            double (*tl)[64] = tile[k % NTB];
            double const *src = B + (size_t)(b0 - 1) * ldb + b0 + lane;
            for (int c0 = q_lo; c0 <= q_hi; c0 += 16) {
                double v[16];
#pragma unroll
                for (int i = 0; i < 16; i++) v[i] = (c0 + i <= q_hi && lane <= q_hi) ? src[(size_t)(c0 + i) * ldb] : 0.0;
#pragma unroll
                for (int i = 0; i < 16; i++) if (c0 + i <= q_hi) tl[c0 + i][lane] = v[i];
            }
        };
        auto flush = [&](int k) {
            int b0, q_lo, q_hi; range(k, b0, q_lo, q_hi);
            double (*tl)[64] = tile[k % NTB];
            double *dst = B + (size_t)(b0 - 1) * ldb + b0 + lane;
            for (int c = q_lo; c <= q_hi + 1; c++) {
                bool const mine = (c == q_lo) ? lane == q_lo : lane <= std::min(c, q_hi);
                if (mine) dst[(size_t)c * ldb] = tl[c][lane];
            }
            int const t = b0 + lane;
            if (lane >= q_lo && lane <= q_hi) { Cc[t] = rot[t_hi - t][0]; Cs[t] = rot[t_hi - t][1]; }
        };
        for (int k = 1; k < K; k++) {                   // (tile 0 came in at the start)
            if (k >= NTB) {
                ht_wait(ctr, 2, k - NTB + 1);
                flush(k - NTB);
            }
            load(k);
            asm volatile("" ::: "memory");
            lds_store(ctr + 4, k + 1);
        }
        for (int k = std::max(0, K - NTB); k < K; k++) {
            ht_wait(ctr, 2, k + 1);
            flush(k);
        }
        if (err && lane == 0 && lds_load(ctr + 12) != 0) *err = 1;     // a wait gave up: the host reports it
        return;
    }

    int const w = wv < G ? wv : G - 1 - wv;             // followers: w = -1, -2, .. (rows below g0)
    int const row = g0 + 64 * w + lane;
    if (w > wtop || g0 + 64 * w + 63 < 0) return;
    bool const rv = row >= 0 && row < g1;
    double y = rv ? B[(size_t)t_hi * ldb + row] : 0.0;

    // ---- follower steps: rotations [clo, chi] of chunk p (block wtop - p/4, from its top) ----
    constexpr int NCH = 64 / CH;
    auto bounds = [&](int p, int &base, int &clo, int &chi) {
        base = g0 + 64 * (wtop - p / NCH) + 64 - CH - CH * (p % NCH);
        chi = std::min(base + CH - 1, t_hi); clo = std::max(base, t_lo);
    };
    auto fetch = [&](double (&xr)[CH], int p, int pend) {
        int base, clo, chi; bounds(p, base, clo, chi);
        double const *src = B + (size_t)(base - 1) * ldb + row;
#pragma unroll
        for (int qq = 0; qq < CH; qq++) {
            int const t = base + qq;
            xr[qq] = (p < pend && rv && t >= clo && t <= chi) ? *src : 0.0;
            src += ldb;
        }
    };
    auto apply = [&](double (&xr)[CH], int p) {
        int base, clo, chi; bounds(p, base, clo, chi);
        if (chi < clo) return;
        int const need = t_hi - clo + 1;
        ht_wait(ctr, 0, need);
        if (!rv) return;
        double *dst = B + (size_t)(base + CH - 1) * ldb + row;
#pragma unroll
        for (int qq = CH - 1; qq >= 0; qq--, dst -= ldb) {
            int const t = base + qq;
            if (t > chi || t < clo) continue;
            double const c = rot[t_hi - t][0], s = rot[t_hi - t][1], x = xr[qq];
            *dst = s * x + c * y;
            y = c * x - s * y;
        }
        if ((DBG & 32) && ts && lane == 0 && w == wtop - p / NCH - 1) ts[64 + (t_hi - chi) / CH] = wall_clock64();
    };
    // a group wave follows the blocks above its own; the other waves follow every block
    int const pend = NCH * ((w >= wbot ? wtop - w : K));
    {
        double xa[CH], xb[CH];
        fetch(xa, 0, pend);
        for (int p = 0; p < pend; p += 2) {
            fetch(xb, p + 1, pend);
            apply(xa, p);
            fetch(xa, p + 2, pend);
            apply(xb, p + 1);
        }
    }
    if (w < wbot) {
        if (rv) B[(size_t)(t_lo - 1) * ldb + row] = y;  // rows above the last rotation: their column t_lo - 1
        return;
    }

    // ---- chain: block k = wtop - w, out of LDS ----
    int const k = wtop - w, b0 = g0 + 64 * w;
    int const q_hi = std::min(63, t_hi - b0), q_lo = std::max(0, t_lo - b0);
    double (*tl)[64] = tile[k % NTB];
    ht_wait(ctr, 1, k + 1);
    // A rolled loop on purpose (64 unrolled steps are 50 KB of code that every chain wave would
    // stream through the instruction cache once), and explicit ds instructions with explicit
    // waits: per rotation one LDS read (next column, consumed one iteration later) and three
    // writes (rotation, result column, counter) that nothing on the chain waits for.  The fill-ins
    // B(t, t-1) -- the tile's diagonal -- are read once into lane t of a register and cleared at the end.
    unsigned const a_tile = lds_addr(&tl[0][0]), a_rot = lds_addr(&rot[0][0]);
    double fv, xn;                                      // (through asm as well: a compiler-tracked LDS read pending at the
                                                        //  loop header would put an s_waitcnt lgkmcnt(0) inside the loop)
    asm volatile("ds_read_b64 %0, %1" : "=v"(fv) : "v"(a_tile + 8u * 65u * lane) : "memory");
    asm volatile("ds_read_b64 %0, %1" : "=v"(xn) : "v"(a_tile + 8u * (64u * q_hi + lane)) : "memory");
    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fv) :: "memory");
#pragma unroll 1
    for (int q = q_hi; q >= q_lo; q--) {
        int const t = b0 + q;
        double x = xn;
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(x) :: "memory");            // the three writes behind the read may stay in flight
        asm volatile("ds_read_b64 %0, %1" : "=v"(xn) : "v"(a_tile + 8u * (64u * std::max(q - 1, q_lo) + lane)) : "memory");
        double const d = readlane_d(y, q), f = readlane_d(fv, q);
        double c, s;
        ht_lartg(d, f, c, s);
        // No select anywhere: in the pivot lane x = f and y = d, so s x + c y IS r; the lanes past the
        // pivot (rows below t) compute and store values that nobody reads -- the loader writes back
        // rows <= t only -- and an `if` around a store would be a compare, an exec save and a branch
        // on the chain (80 ns measured).
        double const yf = s * x + c * y;                                        // column t, final
        y = c * x - s * y;
        asm volatile("ds_write_b128 %0, %1" :: "v"(a_rot + 16u * (t_hi - t)), "v"(v2d{c, s}) : "memory");
        asm volatile("ds_write_b64 %0, %1" :: "v"(a_tile + 8u * (64u * (q + 1) + lane)), "v"(yf) : "memory");
        asm volatile("ds_write_b32 %0, %1" :: "v"(ctr), "v"(t_hi - t + 1) : "memory");   // published: the LDS queue keeps the order
        if ((DBG & 32) && ts && lane == 0 && (t % CH == 0 || q == q_lo)) ts[(t_hi - t) / CH] = wall_clock64();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane >= q_lo && lane <= q_hi) tl[lane][lane] = 0.0;                     // the removed fill-ins
    lds_store(ctr + 8, k + 1);
    if (rv && row < t_lo) B[(size_t)(t_lo - 1) * ldb + row] = y;
}

// max |b_ij|, as the bit pattern of a non-negative double (ordered like integers)
__global__ __launch_bounds__(256) void ht_absmax_kernel(int n, double const *__restrict__ B, int ldb, unsigned long long *out)
{
    int const c = blockIdx.y;
    double m = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) m = fmax(m, fabs(B[(size_t)c * ldb + i]));
    for (int o = 32; o; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

// B <- B * 2^(-dir * e), e = exponent of the largest entry: exact; the sums of squares of the QR step
// and of the chain then stay in range without LAPACK's rescaling branches
__global__ __launch_bounds__(256) void ht_scale_kernel(int n, double *__restrict__ B, int ldb, unsigned long long const *amax, int dir)
{
    int const c = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    double const m = __longlong_as_double((long long)*amax);
    if (i >= n || m == 0.0 || !(m < 1.7e308)) return;
    B[(size_t)c * ldb + i] = scalbn(B[(size_t)c * ldb + i], -dir * ilogb(m));
}

__global__ void ht_clear_lower_kernel(int n, double *__restrict__ B, int ldb)
{
    int const i = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (i < n && i > c) B[(size_t)c * ldb + i] = 0.0;
}

struct HtWorkspace {
    int n = 0, ldp = 0;
    double *Vp = nullptr, *VT = nullptr, *W = nullptr, *G = nullptr, *T = nullptr, *tau = nullptr;
    double *Vp1 = nullptr, *VT1 = nullptr, *Wside = nullptr;    // QR step: second set of panel factors, the side stream's product
    hipEvent_t e_panel[2] = {nullptr, nullptr}, e_pfree[2] = {nullptr, nullptr};
    double *part[2] = {nullptr, nullptr}, *prow[2] = {nullptr, nullptr};
    double *Rc2[2] = {nullptr, nullptr}, *Rs2[2] = {nullptr, nullptr}, *Rc = nullptr, *Rs = nullptr;
    double *Cc2[3] = {nullptr, nullptr, nullptr}, *Cs2[3] = {nullptr, nullptr, nullptr}, *Cc = nullptr, *Cs = nullptr;
    unsigned long long *amax = nullptr;
    double *rp_beta[2] = {nullptr, nullptr}, *rp_up[2] = {nullptr, nullptr}, *rp_alpha[2] = {nullptr, nullptr};   // row pass: [0] B, [1] A
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    hipStream_t main = nullptr, side = nullptr, qstream = nullptr;
    hipEvent_t e_scan = nullptr, e_side = nullptr, e_q[2] = {nullptr, nullptr};   // e_q[0]: last pass over Q, e_q[1]: over Z
    hipEvent_t e_cdone = nullptr;
    std::vector<hipEvent_t> e_chain;                    // one per diagonal group of a sweep
    void ensure(int n_)
    {
        if (!ev[0]) {
            for (auto &e : ev) SN_HIP_CHECK(hipEventCreate(&e));
            int lo = 0, hi = 0;
            SN_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));     // lo = least urgent
            make_stream(&main, true, hi);
            make_stream(&side, false, lo);
            make_stream(&qstream, false, lo);
            for (auto &e : e_q) SN_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            for (auto &e : e_panel) SN_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            for (auto &e : e_pfree) SN_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            SN_HIP_CHECK(hipEventCreateWithFlags(&e_cdone, hipEventDisableTiming));
            SN_HIP_CHECK(hipEventCreateWithFlags(&e_scan, hipEventDisableTiming));
            SN_HIP_CHECK(hipEventCreateWithFlags(&e_side, hipEventDisableTiming));
        }
        while ((int)e_chain.size() < divceil(n_, HGR) + 1) {
            hipEvent_t a;
            SN_HIP_CHECK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
            e_chain.push_back(a);
        }
        if (n_ <= n) return;
        release_buffers();
        n = n_;
        ldp = (int)roundup(n, 16);
        auto alloc = [](double *&p, size_t count) { SN_HIP_CHECK(hipMalloc((void **)&p, count * sizeof(double))); };
        alloc(Vp, (size_t)ldp * QNB); alloc(VT, (size_t)ldp * QNB); alloc(W, (size_t)ldp * QNB);
        alloc(Vp1, (size_t)ldp * QNB); alloc(VT1, (size_t)ldp * QNB); alloc(Wside, (size_t)ldp * QNB);
        alloc(G, QNB * QNB); alloc(T, QNB * QNB); alloc(tau, QNB);
        for (int b = 0; b < 2; b++) { alloc(part[b], (size_t)divceil(n, QROWS) * QP); alloc(prow[b], QNB); }
        for (int b = 0; b < 2; b++) { alloc(Rc2[b], n); alloc(Rs2[b], n); }
        for (int b = 0; b < 3; b++) { alloc(Cc2[b], n); alloc(Cs2[b], n); }
        if (!amax) SN_HIP_CHECK(hipMalloc((void **)&amax, 24));       // [0] max |b|, [1] error flag of the chain kernels, [2] max |a| (two-stage path)
        for (int b = 0; b < 2; b++) {
            alloc(rp_beta[b], (size_t)divceil(n, 64) * n); alloc(rp_up[b], (size_t)divceil(n, 64) * n); alloc(rp_alpha[b], divceil(n, 64));
        }
    }
    void release_buffers()
    {
        double **all[] = {&Vp, &VT, &W, &Vp1, &VT1, &Wside, &G, &T, &tau, &part[0], &part[1], &prow[0], &prow[1], &Rc2[0], &Rc2[1], &Rs2[0], &Rs2[1], &Cc2[0], &Cc2[1], &Cc2[2], &Cs2[0], &Cs2[1], &Cs2[2],
            &rp_beta[0], &rp_beta[1], &rp_up[0], &rp_up[1], &rp_alpha[0], &rp_alpha[1]};
        for (double **p : all) if (*p) { SN_HIP_CHECK(hipFree(*p)); *p = nullptr; }
        n = 0;
    }
};
HtWorkspace g_ht;

void row_pass(hipStream_t s, HtWorkspace &ws, int which, int n, int j, int tri, double *M, int ld)
{
    RowPassArgs a{M, ld, n, j, j + 1, tri, ws.Rc, ws.Rs, ws.rp_beta[which], ws.rp_up[which], ws.rp_alpha[which], (j + 1) / 64};
    int const strips = divceil(n - j - 1, 64), segs = (n - 2) / 64 - a.seg_lo + 1;
    hipLaunchKernelGGL(ht_rowpass1_kernel, dim3(strips, segs), dim3(64), 0, s, a);
    hipLaunchKernelGGL(ht_rowpass2_kernel, dim3(strips), dim3(64), 0, s, a);
    hipLaunchKernelGGL(ht_rowpass3_kernel, dim3(strips, segs), dim3(64), 0, s, a);
}

// B = Q0 R (R in place, strictly lower part cleared), A <- Q0^T A, Q <- Q Q0
void ht_qr_step(hipStream_t s, HtWorkspace &ws, int n, double *dA, int ldA, double *dB, int ldB,
    double *dQ, int ldQ, double *flops)
{
    // Only the trailing columns of B feed the next panel: they stay on `s` behind the panel's column chain.  A and Q
    // -- three quarters of the step's flops, rank-64 updates bound by HBM -- take their updates on the side stream
    // beside the next panels' latency-bound chains, from the second of two sets of panel factors (round 6: the step
    // 0.57 -> 0.18 s at n = 8000 with the split-K tile order of dgemm_mfma.hip and the column kernel above; everything on
    // one stream before).
    // (Round 6, tried: a CU-masked side stream that leaves every fifth CU to the column chain -- its launches wait 30 us
    // instead of 10 for a free CU while the GEMMs of the previous panel run.  One more hardware queue in the process
    // doubled stage 1 of the two-stage path behind it, 1.2 -> 2.3 s at n = 8000: not kept.)
    int const ldp = ws.ldp;
    hipStream_t const q = ws.side;
    int pc = 0;
    for (int p0 = 0; p0 < n; p0 += QNB, pc++) {
        int const nb = std::min(QNB, n - p0), m = n - p0, nwg = divceil(m, QROWS), b = pc & 1;
        double *Vp = b ? ws.Vp1 : ws.Vp, *VT = b ? ws.VT1 : ws.VT;
        if (pc >= 2) SN_HIP_CHECK(hipStreamWaitEvent(s, ws.e_pfree[b], 0));     // the side stream is through with this set
        for (int jj = 0; jj <= nb; jj++)
            hipLaunchKernelGGL(ht_qr_col_kernel, dim3(nwg), dim3(QROWS * QCG), 0, s, n, p0, nb, jj, dB, ldB,
                Vp, ldp, ws.tau, ws.part[(jj + 1) & 1], ws.part[jj & 1], ws.prow[(jj + 1) & 1], ws.prow[jj & 1], nwg);
        dgemm(s, 'T', 'N', nb, nb, m, 1.0, Vp, ldp, Vp, ldp, 0.0, ws.G, QNB);
        hipLaunchKernelGGL(ht_qr_tfactor_kernel, dim3(1), dim3(64), 0, s, nb, ws.G, ws.tau, ws.T);
        dgemm(s, 'N', 'N', m, nb, nb, 1.0, Vp, ldp, ws.T, QNB, 0.0, VT, ldp);
        SN_HIP_CHECK(hipEventRecord(ws.e_panel[b], s));
        int const nc = n - p0 - nb;
        if (nc > 0) {
            double *Bt = dB + (size_t)(p0 + nb) * ldB + p0;
            dgemm(s, 'T', 'N', nb, nc, m, 1.0, VT, ldp, Bt, ldB, 0.0, ws.W, QNB);
            dgemm(s, 'N', 'N', m, nc, nb, -1.0, Vp, ldp, ws.W, QNB, 1.0, Bt, ldB);
            *flops += 4.0 * m * nb * nc;
        }
        SN_HIP_CHECK(hipStreamWaitEvent(q, ws.e_panel[b], 0));
        dgemm(q, 'T', 'N', nb, n, m, 1.0, VT, ldp, dA + p0, ldA, 0.0, ws.Wside, QNB);
        dgemm(q, 'N', 'N', m, n, nb, -1.0, Vp, ldp, ws.Wside, QNB, 1.0, dA + p0, ldA);
        *flops += 4.0 * m * nb * n;
        if (dQ) {
            double *Qt = dQ + (size_t)p0 * ldQ;
            dgemm(q, 'N', 'N', n, nb, m, 1.0, Qt, ldQ, VT, ldp, 0.0, ws.Wside, ldp);
            dgemm(q, 'N', 'T', n, m, nb, -1.0, ws.Wside, ldp, Vp, ldp, 1.0, Qt, ldQ);
            *flops += 4.0 * m * nb * n;
        }
        SN_HIP_CHECK(hipEventRecord(ws.e_pfree[b], q));
    }
    for (int b = 0; b < 2 && b < pc; b++) SN_HIP_CHECK(hipStreamWaitEvent(s, ws.e_pfree[b], 0));
}

} // namespace

void ht_two_stage_release_workspace();
bool ht_two_stage_fits(int n);
int ht_two_stage_device(hipStream_t s, hipStream_t sq, int n, double *A, int lda, double *B, int ldb, double *Q, int ldq,
    double *Z, int ldz, hipEvent_t between);
void hessenberg_triangular_release_workspace() { g_ht.release_buffers(); ht_two_stage_release_workspace(); }

// (dA, dB) general -> (H, T) upper Hessenberg / upper triangular with dQ <- dQ*U1, dZ <- dZ*U2
// (dQ, dZ may be NULL).  stats (may be NULL) is double[8]: [0] total ms, [1] QR step ms, [2] ms of the reduction
// proper (rotation sweeps, or both stages of the two-stage path), [3] executed GEMM flops, [4] rotations (0 on the
// two-stage path), [5] 1 if the two-stage path ran, else 0, [6] its stage 1 ms (else 0), [7] unused.
int hessenberg_triangular_device(hipStream_t caller, int n, double *dA, int ldA, double *dB, int ldB,
    double *dQ, int ldQ, double *dZ, int ldZ, double *stats)
{
    HtWorkspace &ws = g_ht;
    ws.ensure(n);
    double flops = 0.0, rotations = 0.0;
    if (stats) { stats[5] = 0.0; stats[6] = 0.0; stats[7] = 0.0; }
    // the dependent path runs on a stream of the highest priority: its small kernels (one workgroup
    // with 100 KB of LDS) must not queue behind the thousands of workgroups of the streaming passes
    hipStream_t const s = ws.main;
    SN_HIP_CHECK(hipEventRecord(ws.e_cdone, caller));
    SN_HIP_CHECK(hipStreamWaitEvent(s, ws.e_cdone, 0));
    SN_HIP_CHECK(hipEventRecord(ws.ev[0], s));
    SN_HIP_CHECK(hipMemsetAsync(ws.amax, 0, 24, s));
    hipLaunchKernelGGL(ht_absmax_kernel, dim3(std::min(16, divceil(n, 256)), n), dim3(256), 0, s, n, dB, ldB, ws.amax);
    hipLaunchKernelGGL(ht_scale_kernel, dim3(divceil(n, 256), n), dim3(256), 0, s, n, dB, ldB, ws.amax, 1);
    ht_qr_step(s, ws, n, dA, ldA, dB, ldB, dQ, ldQ, &flops);
    hipLaunchKernelGGL(ht_clear_lower_kernel, dim3(divceil(n, 256), n), dim3(256), 0, s, n, dB, ldB);
    SN_HIP_CHECK(hipEventRecord(ws.ev[1], s));

    // Two streams.  Main (the caller's): scan, row pass of B, and per diagonal group the chain and the
    // column pass on the rows of the next two groups -- the dependent path.  Side: row pass of A, the
    // row rotations on Q, and per group the column pass on the remaining rows of B, on A and on Z.
    hipStream_t const side = ws.side;
    static bool attr_set = false;
    if (!attr_set) {
        SN_HIP_CHECK(hipFuncSetAttribute((const void *)ht_colpass_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CL_LDS_BYTES));
        SN_HIP_CHECK(hipFuncSetAttribute((const void *)ht_chain_kernel<HG, HF>, hipFuncAttributeMaxDynamicSharedMemorySize, CHAIN_LDS_DOUBLES(HG) * 8));
        attr_set = true;
    }
    // From n = 1100 on (tuning.h: ht2_min_n) the two-stage Householder reduction (ht_twostage.hip) instead of the rotation sweeps (twice
    // as fast at n = 8000; SN_HT_TWOSTAGE=0 / 1 forces either) -- DESIGN.md section 4d has the measurements
    int const two_stage = ht_two_stage_fits(n) && (tuning().ht_two_stage > 0 || (tuning().ht_two_stage < 0 && n >= tuning().ht2_min_n));
    int two_stage_rc = 0;
    if (two_stage) {
        static hipEvent_t between = nullptr;
        if (!between) SN_HIP_CHECK(hipEventCreate(&between));
        // (Q has taken the QR step's update on `s`: the stream of Q and Z starts behind it)
        SN_HIP_CHECK(hipEventRecord(ws.e_scan, s));
        SN_HIP_CHECK(hipStreamWaitEvent(ws.qstream, ws.e_scan, 0));
        // the Householder kernels of this path take plain sums of squares: A, like B, to max |a| in [1, 2)
        hipLaunchKernelGGL(ht_absmax_kernel, dim3(std::min(16, divceil(n, 256)), n), dim3(256), 0, s, n, dA, ldA, ws.amax + 2);
        hipLaunchKernelGGL(ht_scale_kernel, dim3(divceil(n, 256), n), dim3(256), 0, s, n, dA, ldA, ws.amax + 2, 1);
        // (it refuses a problem before its first launch, never in the middle: on an error the scaling is still undone
        // and the streams are joined below before the call returns)
        two_stage_rc = ht_two_stage_device(s, ws.qstream, n, dA, ldA, dB, ldB, dQ, ldQ, dZ, ldZ, between);
        if (two_stage_rc != 0) SN_HIP_CHECK(hipEventRecord(between, s));
        hipLaunchKernelGGL(ht_scale_kernel, dim3(divceil(n, 256), n), dim3(256), 0, s, n, dA, ldA, ws.amax + 2, -1);
        if (stats) { stats[5] = 1.0; }
        SN_HIP_CHECK(hipEventRecord(ws.e_side, s));
        if (stats) {
            SN_HIP_CHECK(hipEventSynchronize(ws.e_side));
            float t1 = 0.f;
            SN_HIP_CHECK(hipEventElapsedTime(&t1, ws.ev[1], between));
            stats[6] = t1;          // stage 1 ms
        }
    }
    SN_HIP_CHECK(hipEventRecord(ws.e_side, s));
    for (int j = 0; j + 2 < n && !two_stage; j++) {
        SN_HIP_CHECK(hipStreamWaitEvent(s, ws.e_side, 0));          // A, B, Z of the previous sweep complete
        ws.Rc = ws.Rc2[j & 1]; ws.Rs = ws.Rs2[j & 1];               // (Q and Z take the rotations of two sweeps per pass)
        ws.Cc = ws.Cc2[j % 3]; ws.Cs = ws.Cs2[j % 3];               // three buffers: the pass over Z that reads sweeps j-2, j-1 may still run
        if (j >= 1) SN_HIP_CHECK(hipStreamWaitEvent(s, ws.e_q[0], 0));  // the last pass over Q has read both buffers
        hipLaunchKernelGGL(ht_scan_kernel, dim3(1), dim3(1024), 0, s, n, j, dA, ldA, ws.Rc, ws.Rs);
        SN_HIP_CHECK(hipEventRecord(ws.e_scan, s));
        row_pass(s, ws, 0, n, j, 1, dB, ldB);
        SN_HIP_CHECK(hipStreamWaitEvent(side, ws.e_scan, 0));
        row_pass(side, ws, 1, n, j, 0, dA, ldA);
        if (dQ && ((j & 1) || j + 3 >= n)) {
            // Q takes the row rotations of two sweeps per pass (odd j: sweeps j-1 and j); a last even sweep alone
            SN_HIP_CHECK(hipStreamWaitEvent(ws.qstream, ws.e_scan, 0));
            if (j & 1)
                hipLaunchKernelGGL(ht_qpass2_kernel, dim3(divceil(n, 64)), dim3(64), 0, ws.qstream, dQ, ldQ, n, n, j,
                    ws.Rc2[0], ws.Rs2[0], ws.Rc2[1], ws.Rs2[1], -1.0);
            else
                hipLaunchKernelGGL(ht_colpass_kernel, dim3(divceil(n, 64), 1), dim3(64), 0, ws.qstream,
                    ColJob{dQ, ldQ, n}, ColJob{nullptr, 0, 0}, ColJob{nullptr, 0, 0}, n - 1, j + 2, ws.Rc, ws.Rs, -1.0);
            SN_HIP_CHECK(hipEventRecord(ws.e_q[0], ws.qstream));
        }
        if (dZ && (j & 1) && j >= 3) SN_HIP_CHECK(hipStreamWaitEvent(s, ws.e_q[1], 0));   // the pass over Z launched after sweep j-2 read this buffer
        int gi = 0;
        for (int g1 = n; g1 > j + 2; gi++) {
            int const g0 = std::max(0, (g1 - 1) / HGR * HGR);
            // the rows that enter the followers of this launch were updated last by the side stream's
            // column pass of the previous group
            hipLaunchKernelGGL((ht_chain_kernel<HG, HF>), dim3(1), dim3(64 * (HG + HF + 1)), CHAIN_LDS_DOUBLES(HG) * 8, s, n, j, g0, g1, dB, ldB, ws.Cc, ws.Cs, (long long *)nullptr,
                (int *)(ws.amax + 1));
            SN_HIP_CHECK(hipEventRecord(ws.e_chain[gi], s));
            int const t_hi = g1 - 1, t_lo = std::max(g0, j + 2);
            int const near_lo = std::max(0, g0 - 64 * HF);          // rows [near_lo, g0): followers of the chain kernel
            // the rows of B above the followers: next on the chain's own stream (the next launch needs
            // them); A and Z: on the side stream, nobody waits for them before the sweep ends
            if (near_lo > 0)
                hipLaunchKernelGGL(ht_colpass_lds_kernel, dim3(divceil(near_lo, CL_ROWS)), dim3(256), CL_LDS_BYTES, s,
                    dB, ldB, near_lo, t_hi, t_lo, ws.Cc, ws.Cs);
            SN_HIP_CHECK(hipStreamWaitEvent(side, ws.e_chain[gi], 0));
            hipLaunchKernelGGL(ht_colpass_kernel, dim3(divceil(n, 64), 1), dim3(64), 0, side,
                ColJob{dA, ldA, n}, ColJob{nullptr, 0, 0}, ColJob{nullptr, 0, 0}, t_hi, t_lo, ws.Cc, ws.Cs, 1.0);
            g1 = g0;
        }
        SN_HIP_CHECK(hipEventRecord(ws.e_side, side));
        if (dZ && ((j & 1) || j + 3 >= n)) {
            // Z takes the column rotations of two sweeps per pass, like Q the row rotations
            SN_HIP_CHECK(hipEventRecord(ws.e_cdone, s));
            SN_HIP_CHECK(hipStreamWaitEvent(ws.qstream, ws.e_cdone, 0));
            if (j & 1)
                hipLaunchKernelGGL(ht_qpass2_kernel, dim3(divceil(n, 64)), dim3(64), 0, ws.qstream, dZ, ldZ, n, n, j,
                    ws.Cc2[(j - 1) % 3], ws.Cs2[(j - 1) % 3], ws.Cc2[j % 3], ws.Cs2[j % 3], 1.0);
            else
                hipLaunchKernelGGL(ht_colpass_kernel, dim3(divceil(n, 64), 1), dim3(64), 0, ws.qstream,
                    ColJob{dZ, ldZ, n}, ColJob{nullptr, 0, 0}, ColJob{nullptr, 0, 0}, n - 1, j + 2, ws.Cc, ws.Cs, 1.0);
            SN_HIP_CHECK(hipEventRecord(ws.e_q[1], ws.qstream));
        }
        rotations += 2.0 * (n - j - 2);
    }
    SN_HIP_CHECK(hipStreamWaitEvent(s, ws.e_side, 0));
    if (dQ && n > 2 && !two_stage) SN_HIP_CHECK(hipStreamWaitEvent(s, ws.e_q[0], 0));
    if (dZ && n > 2 && !two_stage) SN_HIP_CHECK(hipStreamWaitEvent(s, ws.e_q[1], 0));
    hipLaunchKernelGGL(ht_scale_kernel, dim3(divceil(n, 256), n), dim3(256), 0, s, n, dB, ldB, ws.amax, -1);
    int chain_err = 0;
    SN_HIP_CHECK(hipMemcpyAsync(&chain_err, ws.amax + 1, sizeof(int), hipMemcpyDeviceToHost, s));
    SN_HIP_CHECK(hipEventRecord(ws.ev[2], s));
    SN_HIP_CHECK(hipStreamWaitEvent(caller, ws.ev[2], 0));
    SN_HIP_CHECK(hipStreamSynchronize(s));
    if (stats) {
        float t01 = 0.f, t12 = 0.f;
        SN_HIP_CHECK(hipEventElapsedTime(&t01, ws.ev[0], ws.ev[1]));
        SN_HIP_CHECK(hipEventElapsedTime(&t12, ws.ev[1], ws.ev[2]));
        stats[0] = t01 + t12; stats[1] = t01; stats[2] = t12; stats[3] = flops; stats[4] = rotations;
    }
    if (two_stage_rc != 0) {
        fprintf(stderr, "[starneig-amd] Hessenberg-triangular reduction: the two-stage path refused n = %d (%d); "
            "A and B are not reduced.\n", n, two_stage_rc);
        return STARNEIG_GENERIC_ERROR;
    }
    if (chain_err) {
        fprintf(stderr, "[starneig-amd] Hessenberg-triangular reduction: a wait inside the chain kernel timed out; "
            "the result is not valid.\n");
        return STARNEIG_GENERIC_ERROR;
    }
    return 0;
}

} // namespace sn

#ifdef SN_TEST_HOOKS   // compiled into libstarneig_amd_test.so only (csrc/Makefile), never into the product library
// ---- measurement hook (NOT part of the public C-ABI; scratch/ht_chain_bench.py): average duration
// of one ht_chain_kernel launch on the bottom group (256 rotations) of a random upper Hessenberg B
extern "C" __attribute__((visibility("default")))
double sn_internal_ht_chain_bench(int variant, int reps)
{
    using namespace sn;
    int const n = 2048, ld = 2048;
    double *B0, *B, *Cc, *Cs;
    SN_HIP_CHECK(hipMalloc((void **)&B0, (size_t)ld * n * 8)); SN_HIP_CHECK(hipMalloc((void **)&B, (size_t)ld * n * 8));
    SN_HIP_CHECK(hipMalloc((void **)&Cc, n * 8)); SN_HIP_CHECK(hipMalloc((void **)&Cs, n * 8));
    long long *ts; SN_HIP_CHECK(hipMalloc((void **)&ts, 128 * 8)); SN_HIP_CHECK(hipMemset(ts, 0, 128 * 8));
    lcg_fill(nullptr, n, n, 7u, 1, B0, ld);
    std::vector<double> h((size_t)ld * n);
    SN_HIP_CHECK(hipMemcpy(h.data(), B0, h.size() * 8, hipMemcpyDeviceToHost));
    for (int c = 0; c < n; c++) { for (int r = c + 2; r < n; r++) h[(size_t)c * ld + r] = 0.0; h[(size_t)c * ld + c] += 2.0; }
    SN_HIP_CHECK(hipMemcpy(B0, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; SN_HIP_CHECK(hipEventCreate(&e0)); SN_HIP_CHECK(hipEventCreate(&e1));
    double total = 0.0;
    int const g0 = n - 256, g1 = n;
    for (int r = 0; r < reps + 1; r++) {
        SN_HIP_CHECK(hipMemcpy(B, B0, (size_t)ld * n * 8, hipMemcpyDeviceToDevice));
        SN_HIP_CHECK(hipEventRecord(e0, nullptr));
        auto go = [&](auto kern, int G_, int F_, int gg0) {
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CHAIN_LDS_DOUBLES(G_) * 8));
            hipLaunchKernelGGL(kern, dim3(1), dim3(64 * (G_ + F_ + 1)), CHAIN_LDS_DOUBLES(G_) * 8, nullptr, n, 0, gg0, g1, B, ld, Cc, Cs, ts, (int *)nullptr);
        };
        switch (variant) {
            case 1: go(ht_chain_kernel<4, 0>, 4, 0, g0); break;
            case 2: go(ht_chain_kernel<4, 4>, 4, 4, g0); break;
            case 5: go(ht_chain_kernel<4, 0, 4>, 4, 0, g0); break;
            case 8: go(ht_chain_kernel<1, 0>, 1, 0, n - 64); break;
            case 10: go(ht_chain_kernel<4, 0, 32>, 4, 0, g0); break;
            case 12: go(ht_chain_kernel<8, 0>, 8, 0, n - 512); break;
            default: go(ht_chain_kernel<4, 11>, 4, 11, g0);
        }
        SN_HIP_CHECK(hipEventRecord(e1, nullptr));
        SN_HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0.f; SN_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) total += ms;
    }
    if (variant >= 10) {
        long long h[128];
        SN_HIP_CHECK(hipMemcpy(h, ts, sizeof h, hipMemcpyDeviceToHost));
        printf("chunk: published at / next wave applied at (us after kernel start)\n");
        for (int p = 0; p < 16; p++) printf("  %2d: %7.2f %7.2f\n", p, (h[p] - h[63]) / 100.0, h[64 + p] ? (h[64 + p] - h[63]) / 100.0 : 0.0);
    }
    SN_HIP_CHECK(hipFree(B0)); SN_HIP_CHECK(hipFree(B)); SN_HIP_CHECK(hipFree(Cc)); SN_HIP_CHECK(hipFree(Cs)); SN_HIP_CHECK(hipFree(ts));
    return total / reps * 1e3;
}
#endif  // SN_TEST_HOOKS
