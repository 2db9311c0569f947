// Two-stage reduction of a pencil (A, B), B upper triangular, to Hessenberg-triangular form (the step the
// reference delegates to LAPACK dgghd3, wrappers/lapack.c:143-163) -- from n = 1500 on the product path instead of
// the rotations of hess_tri.hip, whose n^2/2 dependent column rotations bound them.
//
//   Stage 1 (Dackland & Kagstrom; Kagstrom, Kressner, Quintana-Orti, Quintana-Orti 2008): block column by block
//   column (r = 64 wide), bottom up, QR of the (2r x r) blocks of the panel -- reflectors from the left on A, B, Q;
//   the diagonal block of B they fill is restored by r reflectors from the right (RQ of its bottom r rows) on
//   B, A, Z; one RQ of the full r x r block that remains at the top of the block column.  A ends with r
//   sub-diagonals.  Compact-WY factors (V and V T^T from the factorisation kernels), one kernel per application.
//   Stage 2: a Householder bulge chase.  Sweep j, position t: the left reflector of length r that reduces the
//   overhanging column (rows p .. p + r - 1, p = j + 1 + t r), applied to the rows of A, B; the "opposite"
//   reflector from the right whose first column is orthogonal to rows 2 .. r of the r x r block of B (QR of those
//   rows in LDS: no solve with B, singular B included), applied to the columns of B, A.  Sweep j + 1 runs position
//   t while sweep j runs position t + 2: wavefronts of ~n / (2 r) independent steps, three launches each (round 6:
//   the second half of the 75 us factorisation behind the opposite reflectors beside the left application; the near
//   part of the right application; the first half of the NEXT wavefront's factorisations beside the far part of the
//   right application -- scratch/ht2_overlap.py).  Q and Z take the reflectors
//   of 64 sweeps at a time as compact-WY blocks on a second stream -- and so do the rows of A and B ABOVE the first
//   row a group's left reflectors can reach (round 6): those rows only ever see right reflectors again, half of the
//   right application's bytes leave the chase.
// scratch/ht2_proto.py, scratch/ht2_lag.py, scratch/ht2_defer.py and scratch/ht2_overlap.py are the numpy statements of the
// algorithm, of the wavefront order, of the deferred rows and of the order of a wavefront's launches
// (tests/test_ht_twostage_prototype.py).
#include "common.h"
#include "tuning.h"
#include <algorithm>
#include <cmath>
#include <cstdint>
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

// Householder QR of the m x k matrix P in LDS (column-major, leading dimension ldp, k <= 64) by the whole workgroup
// (QT = 1024 threads): R in the upper triangle, the UNSCALED reflector vectors x below the diagonal (v = scl[c] x,
// v_c = 1), tau[0:k], scl[0:k].  LAPACK dlarfg conventions.
//
// History of this routine (profiles/r5_ht_twostage_*.txt): versions 1 and 3 kept the columns in LDS, one column per
// WAVE at a time, with 64-lane shuffle sums (ds_bpermute, six dependent round trips a sum, up to five sums a wave
// and step): 2.7 us per column whether with three barriers and a separate norm pass (1) or two barriers and the
// norm carried along (3) -- 171 us for the 64 x 63 factorisation behind an opposite reflector, 231 us for a 128 x
// 64 panel with its T factor.  Version 2 put every column on ONE lane of one wave (no barrier, no reduction):
// slower still, a single wave hides no LDS latency.  Version 4 below: n = 4000 4.12 s -> 2.63 s, n = 8000
// 13.9 s -> 10.2 s for the whole reduction.

// 16-lane (DPP row) all-reduce of a double: four row rotations, no LDS crossbar round trip
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
    x += ht2_dpp<0x128>(x);   // row_ror:8
    x += ht2_dpp<0x124>(x);   // row_ror:4
    x += ht2_dpp<0x122>(x);   // row_ror:2
    x += ht2_dpp<0x121>(x);   // row_ror:1
    return x;
}

// Column j lives in the REGISTERS of the 16-lane group j of the workgroup (QT / 16 = 64 groups >= k; lane l holds rows l, l + 16, ...,
// MR of them, m <= 16 MR), so every cross-lane sum is four DPP row rotations instead of six ds_bpermute round trips
// and a step touches LDS only for the pivot column.  The group of column c + 1 forms that column's scalars and
// publishes it (unscaled, with R above the diagonal) right after its own update: ONE barrier per column.
template <int MR>
__device__ void group_qr(double *P, int ldp, int m, int k, double *tau, double *scl, int cbeg = 0, int cend = 1 << 30)
{
    // Columns [cbeg, cend) only (stage 2 splits the factorisation behind an opposite reflector over two launches): a
    // call that stops early leaves the columns from cend on in P as they stand, UNpublished; the call that continues
    // publishes column cbeg first.
    __shared__ double piv[2][2];
    int const tid = threadIdx.x, l = tid & 15, g = tid >> 4;
    int const kref = min(m - 1, k), c_lo = min(cbeg, kref), c_hi = min(cend, kref);
    double x[MR];
    #pragma unroll
    for (int r = 0; r < MR; r++) { int const i = l + 16 * r; x[r] = (g < k && i < m) ? P[g * ldp + i] : 0.0; }
    auto publish = [&](int c) {           // by the group of column c, whose updates are complete
        double s = 0.0, alpha = 0.0;
        #pragma unroll
        for (int r = 0; r < MR; r++) { int const i = l + 16 * r; if (i > c) s += x[r] * x[r]; if (i == c) alpha = x[r]; }
        s = row16_sum(s); alpha = row16_sum(alpha);
        double t = 0.0, beta = alpha, scale = 0.0;
        if (s != 0.0) {
            double const nn = alpha * alpha + s;
            if (nn > 1e-280 && nn < 1e280) {
                // the chain of a column ends in these scalars: hardware reciprocal square root and reciprocal with
                // two Newton steps each (a few ulp) instead of the correctly rounded sqrt and two divisions --
                // 120 of the 2100 cycles of a column (scratch/micro/qr_phases.hip)
                double rs = __builtin_amdgcn_rsq(nn);
                rs = rs * (1.5 - 0.5 * nn * rs * rs);
                rs = rs * (1.5 - 0.5 * nn * rs * rs);
                beta = -copysign(nn * rs, alpha);
                t = (beta - alpha) * -copysign(rs, alpha);
                double const dd = alpha - beta;
                double rc = __builtin_amdgcn_rcp(dd);
                rc = rc * (2.0 - dd * rc);
                scale = rc * (2.0 - dd * rc);
            } else {
                beta = -copysign(sqrt(nn), alpha); t = (beta - alpha) / beta; scale = 1.0 / (alpha - beta);
            }
        }
        #pragma unroll
        for (int r = 0; r < MR; r++) { int const i = l + 16 * r; if (i < m) P[c * ldp + i] = (i == c) ? beta : x[r]; }
        if (l == 0) { tau[c] = t; scl[c] = scale; piv[c & 1][0] = t; piv[c & 1][1] = scale; }
    };
    if (g == c_lo && c_lo < kref) publish(c_lo);
    __syncthreads();
    for (int c = c_lo; c < c_hi; c++) {
        double const t = piv[c & 1][0], scale = piv[c & 1][1];
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
            if (g == c + 1 && c + 1 < c_hi) publish(c + 1);
        }
        __syncthreads();
    }
    if (g >= c_hi && g < k) {
        #pragma unroll
        for (int r = 0; r < MR; r++) { int const i = l + 16 * r; if (i < m) P[g * ldp + i] = x[r]; }
        if (l == 0 && g >= kref) { tau[g] = 0.0; scl[g] = 0.0; }
    }
    __syncthreads();
}

typedef double d4v __attribute__((ext_vector_type(4)));
// One 16 x 16 tile D(i, j) = sum_{k < K} a(i, k) b(k, j) on the matrix core (v_mfma_f64_16x16x4_f64), K a multiple of
// 4, operands through accessors (LDS): lane l ends with D(i = (l >> 4) + 4 reg, j = l & 15) in component reg.
template <class FA, class FB>
__device__ __forceinline__ d4v wave_tile(int K, FA a, FB b)
{
    int const l = threadIdx.x & 63, q = l & 15, kk = l >> 4;
    d4v acc = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a(q, k0 + kk), b(k0 + kk, q), acc, 0, 0, 0);
    return acc;
}

// The panel group_qr left (R and the unscaled tails) becomes the explicit V in place: 2 r rows x r columns, unit
// lower trapezoidal, zero beyond m rows and k columns.
__device__ void lds_explicit_v(double *P, int ldp, int m, int k, double const *scl)
{
    for (int e = threadIdx.x; e < 2 * R2 * R2; e += blockDim.x) {
        int const r = e % (2 * R2), c = e / (2 * R2);
        double v = 0.0;
        if (c < k && r < m) v = r > c ? P[c * ldp + r] * scl[c] : (r == c ? 1.0 : 0.0);
        P[c * ldp + r] = v;
    }
    __syncthreads();
}

// T (r x r, upper triangular, leading dimension R2, in LDS) of the compact-WY form H_0 H_1 ... H_{k-1} = I - V T V^T
// (LAPACK dlarft, forward / columnwise) from the explicit V (2 r x r, zero padded; tau[c] = 0 for c >= k).
// G = V^T V on the matrix core; the two 32 x 32 diagonal blocks of T row by row (row i of T depends on row i
// alone: T(i, j) = -tau_j sum_{i <= l < j} T(i, l) G(l, j)), two half waves side by side; the block above the
// diagonal T12 = -T11 (G12 T22) on the matrix core again.  G: r x r scratch, W: 32 x 32 scratch, both in LDS.
// (The first version summed the Gram matrix entry by entry from LDS -- a million reads, 23 us of a 120 us panel --
// and ran all 64 rows of the recurrence through one wave, 9 us.)
__device__ void lds_tfactor(double const *V, int ldv, double const *tau, double *T, double *G, double *W)
{
    int const tid = threadIdx.x, wave = tid >> 6, l = tid & 63, q = l & 15, kk = l >> 4;
    constexpr int H = R2 / 2;
    for (int e = tid; e < R2 * R2; e += blockDim.x) T[e] = 0.0;
    {   // the ten upper 16 x 16 tiles of G, one per wave
        int ti = 0, tj = 0, idx = wave;
        bool mine = false;
        for (int a = 0; a < 4 && !mine; a++) for (int b = a; b < 4; b++) { if (idx == 0) { ti = a; tj = b; mine = true; break; } idx--; }
        if (mine) {
            d4v const d = wave_tile(2 * R2, [&](int i, int k) { return V[(16 * ti + i) * ldv + k]; },
                                            [&](int k, int j) { return V[(16 * tj + j) * ldv + k]; });
            #pragma unroll
            for (int reg = 0; reg < 4; reg++) G[(16 * tj + q) * R2 + 16 * ti + kk + 4 * reg] = d[reg];
        }
    }
    __syncthreads();
    if (tid < R2) {
        int const i = tid, hi = (i / H + 1) * H;           // rows 0 .. 31: columns < 32, rows 32 .. 63: columns < 64
        T[i * R2 + i] = tau[i];
        for (int j = i + 1; j < hi; j++) {
            double g = 0.0;
            for (int c = i; c < j; c++) g += T[c * R2 + i] * G[j * R2 + c];
            T[j * R2 + i] = -tau[j] * g;
        }
    }
    __syncthreads();
    if (wave < 4) {                                          // W = G12 T22
        int const ti = wave & 1, tj = wave >> 1;
        d4v const d = wave_tile(H, [&](int i, int k) { return G[(H + k) * R2 + 16 * ti + i]; },
                                   [&](int k, int j) { return T[(H + 16 * tj + j) * R2 + H + k]; });
        #pragma unroll
        for (int reg = 0; reg < 4; reg++) W[(16 * tj + q) * H + 16 * ti + kk + 4 * reg] = d[reg];
    }
    __syncthreads();
    if (wave < 4) {                                          // T12 = -T11 W
        int const ti = wave & 1, tj = wave >> 1;
        d4v const d = wave_tile(H, [&](int i, int k) { return T[k * R2 + 16 * ti + i]; },
                                   [&](int k, int j) { return W[(16 * tj + j) * H + k]; });
        #pragma unroll
        for (int reg = 0; reg < 4; reg++) T[(H + 16 * tj + q) * R2 + 16 * ti + kk + 4 * reg] = -d[reg];
    }
    __syncthreads();
}

// (V T^T)(row, c) for the 16 x 16 tiles of the 2 r x r product, two per wave of a 1024-thread workgroup, handed to
// `out(row, c, value)`: with it a compact-WY block costs two GEMMs instead of three (X -= (V T^T) (V^T X),
// X -= (X V) (V T^T)^T)
template <class OUT>
__device__ void lds_vtt(double const *V, int ldv, double const *T, OUT out)
{
    int const wave = threadIdx.x >> 6, l = threadIdx.x & 63, q = l & 15, kk = l >> 4;
    for (int tile = wave; tile < (2 * R2 / 16) * (R2 / 16); tile += blockDim.x / 64) {
        int const tr = tile % (2 * R2 / 16), tc = tile / (2 * R2 / 16);
        d4v const d = wave_tile(R2, [&](int i, int k) { return V[k * ldv + 16 * tr + i]; },
                                    [&](int k, int j) { return T[k * R2 + 16 * tc + j]; });
        #pragma unroll
        for (int reg = 0; reg < 4; reg++) out(16 * tr + kk + 4 * reg, 16 * tc + q, d[reg]);
    }
}

// Stage 1, left: QR of the m x nb block X = A(i0:i0+m, jc:jc+nb) (m <= 2 r, nb <= r); R back in place with
// exact zeros below it, V (m x nb, unit lower trapezoidal) and V T^T (m x nb) out, leading dimension ldv both.
__device__ void panel_qr_body(double *__restrict__ X, int ldx, int m, int nb,
    double *__restrict__ V, double *__restrict__ VT, int ldv)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *P = lds, *Tl = P + R2 * LDP, *G = Tl + R2 * R2, *W = G + R2 * R2, *tau = W + R2 * R2 / 4, *scl = tau + R2;
    int const tid = threadIdx.x;
    for (int e = tid; e < m * nb; e += QT) { int const i = e % m, j = e / m; P[j * LDP + i] = X[(size_t)j * ldx + i]; }
    if (tid >= nb && tid < R2) tau[tid] = 0.0;
    __syncthreads();
    group_qr<2 * R2 / 16>(P, LDP, m, nb, tau, scl);
    for (int e = tid; e < m * nb; e += QT) { int const i = e % m, j = e / m; X[(size_t)j * ldx + i] = (i <= j) ? P[j * LDP + i] : 0.0; }
    __syncthreads();
    lds_explicit_v(P, LDP, m, nb, scl);
    lds_tfactor(P, LDP, tau, Tl, G, W);
    for (int e = tid; e < m * nb; e += QT) { int const i = e % m, j = e / m; V[(size_t)j * ldv + i] = P[j * LDP + i]; }
    lds_vtt(P, LDP, Tl, [&](int row, int c, double x) { if (row < m && c < nb) VT[(size_t)c * ldv + row] = x; });
}

// Stage 1, right: the mb x m block Mb = B(i1-mb:i1, i0:i1) (mb <= r, m <= 2 r) becomes [0 R] (R mb x mb upper
// triangular) under G = I - V T V^T from the right: QR of the flipped transpose, flip(Mb^T) = Qr R, G = flip Qr flip.
// V, V T^T: m x mb (rows = columns of the block).
__device__ void rq_body(double *__restrict__ Mb, int ldb, int mb, int m,
    double *__restrict__ V, double *__restrict__ VT, int ldv)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *P = lds, *Tl = P + R2 * LDP, *G = Tl + R2 * R2, *W = G + R2 * R2, *tau = W + R2 * R2 / 4, *scl = tau + R2;
    int const tid = threadIdx.x;
    // P(a, b) = Mb(mb-1-b, m-1-a): m rows, mb columns
    for (int e = tid; e < m * mb; e += QT) {
        int const b = e % mb, a = e / mb;
        P[b * LDP + a] = Mb[(size_t)(m - 1 - a) * ldb + (mb - 1 - b)];
    }
    if (tid >= mb && tid < R2) tau[tid] = 0.0;
    __syncthreads();
    group_qr<2 * R2 / 16>(P, LDP, m, mb, tau, scl);
    // Mb_new(q, c) = R(m-1-c, mb-1-q)
    for (int e = tid; e < m * mb; e += QT) {
        int const b = e % mb, a = e / mb;
        Mb[(size_t)(m - 1 - a) * ldb + (mb - 1 - b)] = (a <= b) ? P[b * LDP + a] : 0.0;
    }
    __syncthreads();
    lds_explicit_v(P, LDP, m, mb, scl);
    lds_tfactor(P, LDP, tau, Tl, G, W);
    // V(row = column index of the block) = flipped rows of Vr
    for (int e = tid; e < m * mb; e += QT) { int const a = e % m, b = e / m; V[(size_t)b * ldv + (m - 1 - a)] = P[b * LDP + a]; }
    lds_vtt(P, LDP, Tl, [&](int row, int c, double x) { if (row < m && c < mb) VT[(size_t)c * ldv + (m - 1 - row)] = x; });
}
constexpr int PANEL_LDS_BYTES = (R2 * LDP + 2 * R2 * R2 + R2 * R2 / 4 + 2 * R2 + 16) * 8;
// Up to two single-workgroup factorisations in one launch (blockIdx.x): the RQ factorisation of a step and the panel
// QR factorisation of the NEXT step of the block column, which depends on nothing the chain of `s` does in between --
// both are ~92 us of latency on one CU each, so the second one is free, and the factors reach their consumers by stream
// order (round 6; before, the panel factorisations ran ahead on a stream of their own and every step paid a
// cross-stream wait, 10-14 us even when the event had long been signalled).
struct FactorJob { int kind; double *X; int ld, a, b; double *V, *VT; };          // kind 0: none, 1: panel QR (m = a, nb = b), 2: RQ (mb = a, m = b)
__global__ __launch_bounds__(QT) void ht2_factor_kernel(FactorJob j0, FactorJob j1)
{
    FactorJob const &j = blockIdx.x == 0 ? j0 : j1;
    if (j.kind == 1) panel_qr_body(j.X, j.ld, j.a, j.b, j.V, j.VT, 2 * R2);
    else if (j.kind == 2) rq_body(j.X, j.ld, j.a, j.b, j.V, j.VT, 2 * R2);
}

// ---- stage 2 --------------------------------------------------------------------------------------------------
// wavefront tau_idx: sweeps jlo .. jlo + count - 1, position t = tau_idx - LAG j.  The reflectors of a step are kept
// until the GROUP of its sweep (GS consecutive sweeps) has gone through: slot (j / GS) mod nslot of the store,
// entry (j mod GS) * tstride + t there.
constexpr int GS = 64;
// Sweep j is at position t = tau - LAG j in wavefront tau.  LAG = 2: the steps (j, t) and (j - 1, t + 2) of one
// wavefront are 127 rows / columns apart (reflectors: 64), and the one entry both touch -- A(p + 127, p + 63), read
// by the older sweep's generation, rewritten by the younger sweep's right application -- is read first, as in the
// sweep-by-sweep order, because generation precedes the applications of a wavefront.  (LAG = 1 is wrong, 3 wasted
// a third of the wavefronts: scratch/ht2_lag.py, tests/test_ht_twostage_prototype.py.)
constexpr int LAG = 2;
struct Wave2 { int n, tau_idx, jlo, count, tstride, nslot; };

__device__ __forceinline__ bool step_of(Wave2 const &w, int k, int &p, int &p1, int &c0, int &ridx)
{
    int const j = w.jlo + k, t = w.tau_idx - LAG * j;
    if (k >= w.count || t < 0 || j > w.n - 3) return false;
    p = j + 1 + t * R2;
    if (p > w.n - 2) return false;
    p1 = min(p + R2, w.n);
    c0 = (t == 0) ? j : p - R2;
    ridx = (((j / GS) % w.nslot) * GS + j % GS) * w.tstride + t;
    return true;
}

// X(p:p1, cols) <- (I - tau v v^T) X for X = A (z = 0) and X = B (z = 1), WITHOUT the step's own diagonal blocks
// A(I, I), B(I, I) (they take H in ht2_near_kernel, see the schedule above the wavefront loop): A: the columns
// [c0 + 1, p) -- chunk 0, at most r - 1 of them -- and [p1, n); B: [p1, n).  16 lanes per column, four rows each (DPP
// sums); a workgroup of 1024 threads takes a chunk of LEFT_CHUNK columns, four per 16-lane group, all in flight
// together.  (The 512-byte pieces of a column start on no particular boundary: 5 lines of 128 bytes for 4 -- in a
// stand-alone copy of this access pattern every mapping tried moved 3.6 TB/s at n = 12000 where an in-place stream
// over as many bytes moves 4.7, scratch/micro/ht2_apply_bw.hip.)
constexpr int LEFT_CHUNK = 128;
__device__ __forceinline__ void apply_left_chunk(Wave2 const &w, int k, int chunk, int z, double *__restrict__ A, int lda,
    double *__restrict__ B, int ldb, double const *__restrict__ HV, double const *__restrict__ HT)
{
    int p, p1, c0, ridx;
    if (!step_of(w, k, p, p1, c0, ridx)) return;
    double const tau = HT[ridx];
    if (tau == 0.0) return;
    bool const isB = z == 1;
    double *X = isB ? B : A;
    int const ld = isB ? ldb : lda, len = p1 - p;
    int cbeg, cend = w.n;
    if (isB) cbeg = p1 + chunk * LEFT_CHUNK;
    else if (chunk == 0) { cbeg = c0 + 1; cend = p; }
    else cbeg = p1 + (chunk - 1) * LEFT_CHUNK;
    if (cbeg >= cend) return;
    int const tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;            // 64 column groups per block
    int const r0 = 4 * l16;
    double v[4];
    #pragma unroll
    for (int q = 0; q < 4; q++) v[q] = (r0 + q < len) ? HV[(size_t)ridx * R2 + r0 + q] : 0.0;
    constexpr int NC = LEFT_CHUNK / 64;
    double y[NC][4], d[NC];
    #pragma unroll
    for (int u = 0; u < NC; u++) {
        int const c = cbeg + grp + 64 * u;
        double const *x = X + (size_t)c * ld + p + r0;
        d[u] = 0.0;
        #pragma unroll
        for (int q = 0; q < 4; q++) { y[u][q] = (c < cend && r0 + q < len) ? x[q] : 0.0; d[u] += v[q] * y[u][q]; }
    }
    #pragma unroll
    for (int u = 0; u < NC; u++) d[u] = row16_sum(d[u]) * tau;
    #pragma unroll
    for (int u = 0; u < NC; u++) {
        int const c = cbeg + grp + 64 * u;
        double *x = X + (size_t)c * ld + p + r0;
        #pragma unroll
        for (int q = 0; q < 4; q++) if (c < cend && r0 + q < len) x[q] = y[u][q] - d[u] * v[q];
    }
}
// (one workgroup per (chunk, step, matrix) of launch M2; nchunk = divceil(n, LEFT_CHUNK) + 1)

// The FAR part of a step's right application: X(top : p - (r - 1), p:p1) <- X (I - tau v v^T) for X = B (z = 0) and
// X = A (z = 1); the rows from p - (r - 1) on are the near part (ht2_near_kernel).  A quarter of a 1024-thread
// workgroup (256 threads, t256) takes 64 rows: wave q of its four the columns 16 q .. 16 q + 15 of the block (lane = row:
// every load is 512 contiguous bytes), the partial row sums meet in LDS.  No thread leaves before the barrier.
// top = (j / GS) GS + 1: the rows above the first row any left reflector of the sweep's GROUP -- or of a
// later one -- can reach.  From the group's first wavefront on they see right reflectors only (a sweep j' of an
// older group is at p' = j' + 1 + (tau - 2 j') r >= top by then), so they take the group's opposite reflectors
// later, as compact-WY blocks beside Z (close_group): the blocks of OLDER groups that overlap one of them in columns
// were generated, and applied to these rows at once, before it; later ones are disjoint from it
// (scratch/ht2_defer.py runs this order in numpy, and the same deferral of the LEFT reflectors' far columns as the
// negative control: a right reflector that straddles the boundary mixes updated and stale columns).
__device__ __forceinline__ void far_tile(Wave2 const &w, int k, int z, int tile, int quarter, int t256,
    double *__restrict__ A, int lda, double *__restrict__ B, int ldb, double const *__restrict__ RV, double const *__restrict__ RT,
    double (*s_d)[4][64])
{
    int p = 0, p1 = 0, c0, ridx = 0;
    bool const ok = step_of(w, k, p, p1, c0, ridx);
    double const tau = ok ? RT[ridx] : 0.0;
    int const len = p1 - p;
    double *X = z == 0 ? B : A;
    int const ld = z == 0 ? ldb : lda;
    // (row tiles start on a multiple of 16 rows at or below top: 128-byte lines; the rows below top are masked)
    int const top = ((w.jlo + k) / GS) * GS + 1, base = top & ~15;
    int const rows = max(top, p - (R2 - 1));
    // (the wave's column offset made uniform for the compiler: the reflector entries are scalar loads, not 32 more
    // vector registers under the launch's 64)
    int const lane = t256 & 63, part = __builtin_amdgcn_readfirstlane(t256 >> 6);
    int const row = base + tile * 64 + lane, q0 = 16 * part;
    bool const live = ok && tau != 0.0 && row < rows && row >= top;
    double *x = X + (size_t)(p + q0) * ld + row;
    double y[16], v[16], d = 0.0;
    #pragma unroll
    for (int q = 0; q < 16; q++) {
        v[q] = (ok && q0 + q < len) ? RV[(size_t)ridx * R2 + q0 + q] : 0.0;
        y[q] = (live && q0 + q < len) ? x[(size_t)q * ld] : 0.0;
        d += y[q] * v[q];
    }
    s_d[quarter][part][lane] = d;
    __syncthreads();
    d = ((s_d[quarter][0][lane] + s_d[quarter][1][lane]) + (s_d[quarter][2][lane] + s_d[quarter][3][lane])) * tau;
    #pragma unroll
    for (int q = 0; q < 16; q++) if (live && q0 + q < len) x[(size_t)q * ld] = y[q] - d * v[q];
}

// The reflectors of a step.  The left reflector H = I - th v v^T comes from A's overhanging column (v, th -> the step's
// entry of HV, HT; the column is reduced in place).  The opposite reflector: M = H Bb in LDS for the step's block
// Bb = B(p:p1, p:p1) as it stands BEFORE H, x orthogonal to rows 1 .. len-1 of M (QR of those rows, transposed;
// x = the last column of the full Q), G = I - tz w w^T with G e_1 = +-x (w, tz -> GV, GT).  Bb is a full block (the
// bulge of B travels with the sweep; a step restores its first column only), so there is no triangular short cut
// to x: 75 us of latency, one workgroup a step on at most n / 127 of the 256 CUs.  They are split over TWO launches so
// that they hide under both HBM-bound applications of the chase (the schedule above the wavefront loop):
// gen_first -- H, M, the first GEN_SPLIT columns of the factorisation; its state (the panel, tau, scl) to `state` --
// beside the far part of the previous wavefront's right application, gen_second -- the rest -- beside the left
// application of its own wavefront.
constexpr int GEN_LQ = R2 + 1;
constexpr int GEN_STATE = R2 * GEN_LQ + 2 * R2;       // doubles a step: the panel, tau, scl
constexpr int GEN_SPLIT = 27;                         // (the early columns are the long ones: 27 of 63 are half of the time)
// A wavefront whose left application alone outlasts the whole factorisation keeps it in the second launch (split 0:
// gen_first forms H and M only) -- beside the shorter far part its workgroups only take CUs from it.
constexpr double GEN_HIDE_US = 100.0, LEFT_BYTES_PER_US = 3.8e6;
__device__ void gen_first(Wave2 const &w, int k, int split, double *__restrict__ A, int lda, double const *__restrict__ B, int ldb,
    double *__restrict__ HV, double *__restrict__ HT, double *__restrict__ state)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int LQ = GEN_LQ;
    double *P = lds, *Bs = P + R2 * LQ, *tau = Bs + R2 * LQ, *scl = tau + R2, *s_v = scl + R2, *s_u = s_v + R2;
    __shared__ double s_th;
    int const tid = threadIdx.x;
    int p, p1, c0, ridx;
    if (!step_of(w, k, p, p1, c0, ridx)) return;
    int const len = p1 - p, kq = len - 1;
    if (tid < 64) {
        int const lane = tid;
        double *col = A + (size_t)c0 * lda + p;
        double const x = lane < len ? col[lane] : 0.0;
        double const ss = wsum64(lane >= 1 ? x * x : 0.0);
        double const alpha = __shfl(x, 0);
        double t = 0.0, beta = alpha, scale = 0.0;
        if (ss != 0.0) { beta = -copysign(sqrt(alpha * alpha + ss), alpha); t = (beta - alpha) / beta; scale = 1.0 / (alpha - beta); }
        double const v = lane == 0 ? 1.0 : (lane < len ? x * scale : 0.0);
        HV[(size_t)ridx * R2 + lane] = v;
        s_v[lane] = v;
        if (lane == 0) { HT[ridx] = t; s_th = t; }
        if (lane < len) col[lane] = lane == 0 ? beta : 0.0;
    }
    for (int idx = tid; idx < len * len; idx += QT) {
        int const i = idx % len, j = idx / len;
        Bs[j * LQ + i] = B[(size_t)(p + j) * ldb + p + i];
    }
    __syncthreads();
    if (tid < len) {                       // u = v^T Bb
        double u = 0.0;
        for (int i = 0; i < len; i++) u += s_v[i] * Bs[tid * LQ + i];
        s_u[tid] = u * s_th;
    }
    __syncthreads();
    // P(a, b) = M(b + 1, a) = Bb(b + 1, a) - th v(b + 1) u(a): len rows, len - 1 columns
    for (int idx = tid; idx < len * kq; idx += QT) {
        int const a = idx % len, b = idx / len;
        P[b * LQ + a] = Bs[a * LQ + b + 1] - s_v[b + 1] * s_u[a];
    }
    __syncthreads();
    group_qr<R2 / 16>(P, LQ, len, kq, tau, scl, 0, split);
    double *st = state + (size_t)k * GEN_STATE;
    for (int idx = tid; idx < R2 * LQ + 2 * R2; idx += QT)
        st[idx] = idx < R2 * LQ ? P[idx] : (idx < R2 * LQ + R2 ? tau[idx - R2 * LQ] : scl[idx - R2 * LQ - R2]);
}
__device__ void gen_second(Wave2 const &w, int k, int split, double const *__restrict__ state, double *__restrict__ GV, double *__restrict__ GT)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int LQ = GEN_LQ;
    double *P = lds, *Bs = P + R2 * LQ, *tau = Bs + R2 * LQ, *scl = tau + R2;
    int const tid = threadIdx.x;
    int p, p1, c0, ridx;
    if (!step_of(w, k, p, p1, c0, ridx)) return;
    GV += (size_t)ridx * R2; GT += ridx;
    int const len = p1 - p, kq = len - 1;
    double const *st = state + (size_t)k * GEN_STATE;
    for (int idx = tid; idx < R2 * LQ + 2 * R2; idx += QT) {
        double const x = st[idx];
        if (idx < R2 * LQ) P[idx] = x; else if (idx < R2 * LQ + R2) tau[idx - R2 * LQ] = x; else scl[idx - R2 * LQ - R2] = x;
    }
    __syncthreads();
    group_qr<R2 / 16>(P, LQ, len, kq, tau, scl, split);
    if (tid < 16) {
        // e = H_0 ... H_{kq-1} e_{len-1}, four entries per lane of ONE 16-lane group (DPP sums); then the
        // reflector from x = e (dlarfg)
        int const l = tid;
        double ev[4], vi[4];
        #pragma unroll
        for (int r = 0; r < 4; r++) ev[r] = (l + 16 * r == len - 1) ? 1.0 : 0.0;
        for (int c = kq - 1; c >= 0; c--) {
            double const t = tau[c];
            if (t == 0.0) continue;
            double const sc = scl[c];
            double d = 0.0;
            #pragma unroll
            for (int r = 0; r < 4; r++) {
                int const i = l + 16 * r;
                vi[r] = (i > c && i < len) ? sc * P[c * LQ + i] : (i == c ? 1.0 : 0.0);
                d += vi[r] * ev[r];
            }
            d = row16_sum(d) * t;
            #pragma unroll
            for (int r = 0; r < 4; r++) ev[r] -= d * vi[r];
        }
        double ss = 0.0, alpha = 0.0;
        #pragma unroll
        for (int r = 0; r < 4; r++) {
            int const i = l + 16 * r;
            if (i >= len) ev[r] = 0.0;
            if (i >= 1) ss += ev[r] * ev[r]; else alpha = ev[r];
        }
        ss = row16_sum(ss); alpha = row16_sum(alpha);
        double t = 0.0, scale = 0.0;
        if (ss != 0.0) { double const beta = -copysign(sqrt(alpha * alpha + ss), alpha); t = (beta - alpha) / beta; scale = 1.0 / (alpha - beta); }
        #pragma unroll
        for (int r = 0; r < 4; r++) {
            int const i = l + 16 * r;
            GV[i] = i == 0 ? 1.0 : (i < len ? ev[r] * scale : 0.0);
        }
        if (l == 0) GT[0] = t;
    }
}
constexpr int GEN_LDS_BYTES = (2 * R2 * (R2 + 1) + 4 * R2 + 16) * 8;

// Launch M1 of a wavefront: the first halves of the NEXT wavefront's generations (wn, one workgroup a step, first in
// the grid) beside the far parts of THIS wavefront's right applications (wc: ntile4 workgroups of four row tiles a
// step and matrix).  Launch M2: the second halves of a wavefront's generations beside its left applications.
__global__ __launch_bounds__(QT, 8) void ht2_m1_kernel(Wave2 wn, int split, Wave2 wc, int ntile4, double *__restrict__ A, int lda, double *__restrict__ B, int ldb,
    double *__restrict__ HV, double *__restrict__ HT, double const *__restrict__ GV, double const *__restrict__ GT, double *__restrict__ state)
{
    __shared__ double s_d[4][4][64];
    if ((int)blockIdx.x < wn.count) { gen_first(wn, blockIdx.x, split, A, lda, B, ldb, HV, HT, state); return; }
    int const idx = blockIdx.x - wn.count, tile4 = idx % ntile4, rest = idx / ntile4;
    int const quarter = threadIdx.x >> 8;
    far_tile(wc, rest % wc.count, rest / wc.count, 4 * tile4 + quarter, quarter, threadIdx.x & 255, A, lda, B, ldb, GV, GT, s_d);
}
__global__ __launch_bounds__(QT, 8) void ht2_m2_kernel(Wave2 w, int split, int nchunk, double *__restrict__ A, int lda, double *__restrict__ B, int ldb,
    double const *__restrict__ HV, double const *__restrict__ HT, double *__restrict__ GV, double *__restrict__ GT, double const *__restrict__ state)
{
    if ((int)blockIdx.x < w.count) { gen_second(w, blockIdx.x, split, state, GV, GT); return; }
    int const idx = blockIdx.x - w.count, chunk = idx % nchunk, rest = idx / nchunk;
    apply_left_chunk(w, rest % w.count, chunk, rest / w.count, A, lda, B, ldb, HV, HT);
}

// The NEAR part of a step's right application, and before it the left reflector on the step's own diagonal blocks:
//   tile 1 (rows I = [p, p1)):  X(I, I) <- H X(I, I), then X(I, I) <- X(I, I) G          (X = B: blockIdx.z 0, X = A: 1)
//   tile 0 (rows [p - (r - 1), p), never above the group's top):  X(rows, I) <- X(rows, I) G
//   tile 2 (A only, rows [p1, min(p1 + r, n))):                   A(rows, I) <- A(rows, I) G
// and B(p + 1 : p1, p) = 0 exactly.  These are the entries the NEXT wavefront's reflectors are generated from
// (A(p + r : p + 2r, p) for the sweep's own next left reflector, B(p - r + 1 : p + 1, p) as the last column of the
// younger neighbour's next block) -- everything else of the right application is the far part
// (ht2_apply_right_kernel, far = 1) and waits on the second stream.  256 threads: lane = row of the tile (512
// contiguous bytes a load; the tiles start at p, not on a 128-byte boundary -- 3 x 32 KB a step), wave q = the columns
// p + 16 q .. p + 16 q + 15; the column sums of H are full-wave sums (DPP rows + two shuffles), the row sums of G meet
// in LDS.
__device__ __forceinline__ double near_wave_sum(double x)
{
    x = row16_sum(x);
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
}
__global__ __launch_bounds__(256) void ht2_near_kernel(Wave2 w, double *__restrict__ A, int lda, double *__restrict__ B, int ldb,
    double const *__restrict__ HV, double const *__restrict__ HT, double const *__restrict__ GV, double const *__restrict__ GT)
{
    __shared__ double s_d[4][64];
    int const tile = blockIdx.x, k = blockIdx.y, z = blockIdx.z;
    int p, p1, c0, ridx;
    if (!step_of(w, k, p, p1, c0, ridx)) return;
    if (z == 0 && tile == 2) return;
    int const len = p1 - p;
    double *X = z == 0 ? B : A;
    int const ld = z == 0 ? ldb : lda;
    int const top = ((w.jlo + k) / GS) * GS + 1, cut = max(top, p - (R2 - 1));
    int const lane = threadIdx.x & 63, wq = threadIdx.x >> 6, q0 = 16 * wq;
    int const row = tile == 0 ? p - R2 + lane : (tile == 1 ? p + lane : p1 + lane);
    int const hi = tile == 0 ? p : (tile == 1 ? p1 : min(p1 + R2, w.n)), lo = tile == 0 ? cut : (tile == 1 ? p : p1);
    bool const live = row >= lo && row < hi;
    double *x = X + (size_t)(p + q0) * ld + row;
    double y[16], g[16];
    #pragma unroll
    for (int q = 0; q < 16; q++) {
        g[q] = q0 + q < len ? GV[(size_t)ridx * R2 + q0 + q] : 0.0;
        y[q] = (live && q0 + q < len) ? x[(size_t)q * ld] : 0.0;
    }
    if (tile == 1) {
        double const th = HT[ridx];
        if (th != 0.0) {
            double const v = lane < len ? HV[(size_t)ridx * R2 + lane] : 0.0;
            #pragma unroll
            for (int q = 0; q < 16; q++) y[q] -= (th * near_wave_sum(v * y[q])) * v;
        }
    }
    double const tz = GT[ridx];
    double d = 0.0;
    #pragma unroll
    for (int q = 0; q < 16; q++) d += y[q] * g[q];
    s_d[wq][lane] = d;
    __syncthreads();
    d = ((s_d[0][lane] + s_d[1][lane]) + (s_d[2][lane] + s_d[3][lane])) * tz;
    #pragma unroll
    for (int q = 0; q < 16; q++) if (live && q0 + q < len) x[(size_t)q * ld] = y[q] - d * g[q];
    if (z == 0 && tile == 1 && wq == 0 && live && row > p) x[0] = 0.0;            // B(p+1:p1, p) = 0 exactly
}

// Q and Z do not take part in the chase: the reflectors of a group of GS sweeps are applied to them once the group
// is through, position by position as compact-WY blocks.  The reflectors (j0 + jj, t), jj = 0 .. k-1, of position t
// act on the columns col0 + jj .. col0 + jj + len - 1 (col0 = j0 + 1 + 64 t): V is a (k - 1 + 64) x k parallelogram.
// Blocks of one group go in DEcreasing t: reflector (j, t + 1) overlaps (j', t) in one column exactly when j' > j,
// and the chase applies it first; all other pairs of different positions are disjoint.  This kernel: V and V T^T
// (leading dimension 2 r, zero filled) of every position of the group, for the left reflectors
// (blockIdx.y = 0, for Q) and the opposite ones (1, for Z).
constexpr int LDVS = 2 * R2 + 1;
__global__ __launch_bounds__(QT) void ht2_group_wy_kernel(int n, int j0, int gsize, int tstride, int slot,
    double const *__restrict__ HV, double const *__restrict__ HT, double const *__restrict__ GV, double const *__restrict__ GT,
    double *__restrict__ Vb, double *__restrict__ VTb)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *V = lds, *G = V + R2 * LDVS, *T = G + R2 * R2, *W = T + R2 * R2, *tau = W + R2 * R2 / 4;
    int const t = blockIdx.x, which = blockIdx.y, tid = threadIdx.x;
    int const col0 = j0 + 1 + R2 * t;
    int const k = min(gsize, n - 1 - col0);                 // sweeps with p = col0 + jj <= n - 2
    if (k <= 0) return;
    double const *RV = which ? GV : HV, *RT = which ? GT : HT;
    for (int e = tid; e < R2 * LDVS; e += QT) V[e] = 0.0;
    if (tid < R2) tau[tid] = 0.0;
    __syncthreads();
    for (int e = tid; e < k * R2; e += QT) {
        int const jj = e / R2, i = e % R2;
        size_t const ridx = (size_t)(slot * GS + jj) * tstride + t;
        int const len = min(R2, n - (col0 + jj));
        if (i < len) V[jj * LDVS + jj + i] = RV[ridx * R2 + i];
        if (i == 0) tau[jj] = RT[ridx];
    }
    __syncthreads();
    lds_tfactor(V, LDVS, tau, T, G, W);
    double *Vo = Vb + ((size_t)which * tstride + t) * (2 * R2 * R2), *VTo = VTb + ((size_t)which * tstride + t) * (2 * R2 * R2);
    for (int e = tid; e < k * 2 * R2; e += QT) { int const r = e % (2 * R2), jj = e / (2 * R2); Vo[(size_t)jj * 2 * R2 + r] = V[jj * LDVS + r]; }
    lds_vtt(V, LDVS, T, [&](int row, int c, double x) { if (c < k) VTo[(size_t)c * 2 * R2 + row] = x; });
}
constexpr int GROUP_LDS_BYTES = (R2 * LDVS + 2 * R2 * R2 + R2 * R2 / 4 + R2 + 16) * 8;

// ---- compact-WY applications: one kernel for up to two targets -----------------------------------------------
// Stage 1 is bound by the HOST: 24 runtime calls a step (kernel launches, event records and waits) at 9 us each are
// its 1.64 s at n = 8000 (13 us each once a QZ run has left its streams and events in the process: 2.24 s).  So the
// two GEMM launches of an application (W = V^T X, X -= (V T^T) W) are one kernel -- a workgroup keeps its slab of X in
// LDS, forms W there and writes the slab once; V, then V T^T, through the same LDS buffer; the products on the matrix
// core -- and the two matrices that take the same factors go in one launch (blockIdx.y).
constexpr int WY_T = 512, WY_SLAB = 32;
constexpr int WY_LDV = 2 * R2 + 1, WY_LDW = R2 + 1, WY_LDR = WY_SLAB + 1;
constexpr int WY_LEFT_LDS = (R2 * WY_LDV + WY_SLAB * WY_LDV + WY_SLAB * WY_LDW) * 8;
constexpr int WY_RIGHT_LDS = (R2 * WY_LDV + 2 * R2 * WY_LDR + R2 * WY_LDR) * 8;
struct WyTargets { double *X[3]; int ld[3]; int extent[3]; };      // extent: columns (left) / rows (right)

// a factor (2 r x r, leading dimension 2 r, zero beyond m rows and k columns) into registers -- every load in flight
// at once, and V T^T on its way while W is still being formed from V -- and from there into LDS
constexpr int WY_FREG = 2 * R2 * R2 / WY_T;
__device__ __forceinline__ void wy_fetch_factor(double (&f)[WY_FREG], double const *__restrict__ V, int m, int k)
{
    #pragma unroll
    for (int u = 0; u < WY_FREG; u++) {
        int const e = threadIdx.x + u * WY_T, r = e % (2 * R2), c = e / (2 * R2);
        f[u] = (r < m && c < k) ? V[(size_t)c * 2 * R2 + r] : 0.0;
    }
}
__device__ __forceinline__ void wy_store_factor(double *Vs, double const (&f)[WY_FREG])
{
    #pragma unroll
    for (int u = 0; u < WY_FREG; u++) {
        int const e = threadIdx.x + u * WY_T, r = e % (2 * R2), c = e / (2 * R2);
        Vs[c * WY_LDV + r] = f[u];
    }
}

// X (m x ncols, m <= 2 r) <- X - (V T^T) (V^T X); blockIdx.x: slab of WY_SLAB columns, blockIdx.y: target
__global__ __launch_bounds__(WY_T) void ht2_wy_left_kernel(double const *__restrict__ V, double const *__restrict__ VT, int m, int k,
    WyTargets tg)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *Vs = lds, *Xs = Vs + R2 * WY_LDV, *Ws = Xs + WY_SLAB * WY_LDV;
    int const tid = threadIdx.x, wave = tid >> 6, l = tid & 63, q = l & 15, kk = l >> 4, z = blockIdx.y;
    int const ncols = tg.extent[z], ldx = tg.ld[z], c0 = blockIdx.x * WY_SLAB;
    if (c0 >= ncols) return;
    double *X = tg.X[z];
    int const nc = min(WY_SLAB, ncols - c0);
    double f[WY_FREG], xr[WY_SLAB * 2 * R2 / WY_T];
    wy_fetch_factor(f, V, m, k);
    #pragma unroll
    for (int u = 0; u < WY_SLAB * 2 * R2 / WY_T; u++) {
        int const e = tid + u * WY_T, r = e % (2 * R2), c = e / (2 * R2);
        xr[u] = (r < m && c < nc) ? X[(size_t)(c0 + c) * ldx + r] : 0.0;
    }
    wy_store_factor(Vs, f);
    #pragma unroll
    for (int u = 0; u < WY_SLAB * 2 * R2 / WY_T; u++) { int const e = tid + u * WY_T; Xs[(e / (2 * R2)) * WY_LDV + e % (2 * R2)] = xr[u]; }
    wy_fetch_factor(f, VT, m, k);
    __syncthreads();
    {   // W (r x slab) = V^T X: 4 x 2 tiles, one per wave
        int const ti = wave & 3, tj = wave >> 2;
        d4v const d = wave_tile(2 * R2, [&](int i, int kr) { return Vs[(16 * ti + i) * WY_LDV + kr]; },
                                        [&](int kr, int j) { return Xs[(16 * tj + j) * WY_LDV + kr]; });
        #pragma unroll
        for (int reg = 0; reg < 4; reg++) Ws[(16 * tj + q) * WY_LDW + 16 * ti + kk + 4 * reg] = d[reg];
    }
    __syncthreads();
    wy_store_factor(Vs, f);
    __syncthreads();
    // X -= (V T^T) W: 8 x 2 tiles, two per wave
    for (int tile = wave; tile < 16; tile += WY_T / 64) {
        int const ti = tile & 7, tj = tile >> 3;
        d4v const d = wave_tile(R2, [&](int i, int kr) { return Vs[kr * WY_LDV + 16 * ti + i]; },
                                    [&](int kr, int j) { return Ws[(16 * tj + j) * WY_LDW + kr]; });
        #pragma unroll
        for (int reg = 0; reg < 4; reg++) Xs[(16 * tj + q) * WY_LDV + 16 * ti + kk + 4 * reg] -= d[reg];
    }
    __syncthreads();
    for (int e = tid; e < WY_SLAB * 2 * R2; e += WY_T) {
        int const r = e % (2 * R2), c = e / (2 * R2);
        if (r < m && c < nc) X[(size_t)(c0 + c) * ldx + r] = Xs[c * WY_LDV + r];
    }
}

// X (nrows x m, m <= 2 r) <- X - (X V) (V T^T)^T; blockIdx.x: slab of WY_SLAB rows
__device__ __forceinline__ void wy_right_body(double const *__restrict__ V, double const *__restrict__ VT, int m, int k,
    double *__restrict__ X, int ldx, int nrows)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *Vs = lds, *Xs = Vs + R2 * WY_LDV, *Ws = Xs + 2 * R2 * WY_LDR;       // Xs[c][r], Ws[kk][r]
    int const tid = threadIdx.x, wave = tid >> 6, l = tid & 63, q = l & 15, kk = l >> 4;
    int const r0 = blockIdx.x * WY_SLAB;
    if (r0 >= nrows) return;
    int const nr = min(WY_SLAB, nrows - r0);
    double f[WY_FREG], xr[2 * R2 * WY_SLAB / WY_T];
    wy_fetch_factor(f, V, m, k);
    #pragma unroll
    for (int u = 0; u < 2 * R2 * WY_SLAB / WY_T; u++) {
        int const e = tid + u * WY_T, r = e % WY_SLAB, c = e / WY_SLAB;
        xr[u] = (r < nr && c < m) ? X[(size_t)c * ldx + r0 + r] : 0.0;
    }
    wy_store_factor(Vs, f);
    #pragma unroll
    for (int u = 0; u < 2 * R2 * WY_SLAB / WY_T; u++) { int const e = tid + u * WY_T; Xs[(e / WY_SLAB) * WY_LDR + e % WY_SLAB] = xr[u]; }
    wy_fetch_factor(f, VT, m, k);
    __syncthreads();
    {   // W (slab x r) = X V: 2 x 4 tiles, one per wave
        int const ti = wave & 1, tj = wave >> 1;
        d4v const d = wave_tile(2 * R2, [&](int i, int kr) { return Xs[kr * WY_LDR + 16 * ti + i]; },
                                        [&](int kr, int j) { return Vs[(16 * tj + j) * WY_LDV + kr]; });
        #pragma unroll
        for (int reg = 0; reg < 4; reg++) Ws[(16 * tj + q) * WY_LDR + 16 * ti + kk + 4 * reg] = d[reg];
    }
    __syncthreads();
    wy_store_factor(Vs, f);
    __syncthreads();
    // X -= W (V T^T)^T: 2 x 8 tiles, two per wave
    for (int tile = wave; tile < 16; tile += WY_T / 64) {
        int const ti = tile & 1, tj = tile >> 1;
        d4v const d = wave_tile(R2, [&](int i, int kr) { return Ws[kr * WY_LDR + 16 * ti + i]; },
                                    [&](int kr, int j) { return Vs[kr * WY_LDV + 16 * tj + j]; });
        #pragma unroll
        for (int reg = 0; reg < 4; reg++) Xs[(16 * tj + q) * WY_LDR + 16 * ti + kk + 4 * reg] -= d[reg];
    }
    __syncthreads();
    for (int e = tid; e < 2 * R2 * WY_SLAB; e += WY_T) {
        int const r = e % WY_SLAB, c = e / WY_SLAB;
        if (r < nr && c < m) X[(size_t)c * ldx + r0 + r] = Xs[c * WY_LDR + r];
    }
}
// one factor, up to three targets (blockIdx.y)
__global__ __launch_bounds__(WY_T) void ht2_wy_right_kernel(double const *__restrict__ V, double const *__restrict__ VT, int m, int k,
    WyTargets tg)
{
    int const z = blockIdx.y;
    wy_right_body(V, VT, m, k, tg.X[z], tg.ld[z], tg.extent[z]);
}
// two (factor, target) pairs in one launch (blockIdx.y): stage 1 applies a step's left factor to Q and its right factor
// to Z together -- the host, not the GPU, bounds stage 1, and a launch less is a runtime call less
struct WyJob { double const *V, *VT; int m, k; double *X; int ld, nrows; };
__global__ __launch_bounds__(WY_T) void ht2_wy_right2_kernel(WyJob j0, WyJob j1)
{
    WyJob const &j = blockIdx.y == 0 ? j0 : j1;
    wy_right_body(j.V, j.VT, j.m, j.k, j.X, j.ld, j.nrows);
}

constexpr int RING = 8;             // stage 1: factor slots in flight between the critical stream and the stream of Q and Z
constexpr int MAXSLOT = 256;        // stage 2: groups of sweeps whose reflectors are kept at a time
constexpr int EXTRASLOT = 40;        // ... beyond what the chase itself needs: a group's reflectors stay until the side stream has applied them to Q, Z and
                                    // the top rows, and with the bare minimum (4 at n = 12000) the chase waited for that stream (stage 2 6.07 -> 5.95 s)
inline int ht2_tstride(int n) { return (n - 3) / R2 + 1; }
// a group is in the chase for LAG (GS - 1) + tstride wavefronts, the next one starts LAG GS wavefronts after it; one
// more slot for the group whose blocks the second stream is still applying
inline int ht2_need_slots(int n) { return (ht2_tstride(n) + LAG * GS - LAG) / (LAG * GS) + 2; }
inline int ht2_nslot(int n) { return std::min(MAXSLOT, ht2_need_slots(n) + EXTRASLOT); }
struct Ht2Workspace {
    int n = 0;
    double *V = nullptr, *VT = nullptr;             // stage 1: two rings (QR, RQ) of RING slots of V and V T^T (2r x r each)
    double *HV = nullptr, *HT = nullptr, *GV = nullptr, *GT = nullptr;      // stage 2: nslot groups of GS x tstride reflectors
    double *Vb = nullptr, *VTb = nullptr;           // stage 2: the compact-WY blocks of the group being applied (2 x tstride)
    double *gstate = nullptr;                       // stage 2: the half-finished factorisations between the two launches of a generation (GEN_STATE doubles a step)
    int maxcount = 0;
    bool attr = false;
    hipStream_t pstream = nullptr;                  // stage 1: the panel factorisations run ahead on it
    hipEvent_t ready[RING] = {}, used[RING] = {}, used_s[RING] = {}, ready_r[RING] = {}, used_r[RING] = {}, column = nullptr, tail = nullptr;
    hipEvent_t through[MAXSLOT] = {}, applied[MAXSLOT] = {};
    void ensure(int n_)
    {
        if (!attr) {
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)ht2_factor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PANEL_LDS_BYTES));
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)ht2_m1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GEN_LDS_BYTES));
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)ht2_m2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GEN_LDS_BYTES));
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)ht2_group_wy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GROUP_LDS_BYTES));
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)ht2_wy_left_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WY_LEFT_LDS));
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)ht2_wy_right_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WY_RIGHT_LDS));
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)ht2_wy_right2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WY_RIGHT_LDS));
            for (int k = 0; k < RING; k++) {
                SN_HIP_CHECK(hipEventCreateWithFlags(&ready[k], hipEventDisableTiming));
                SN_HIP_CHECK(hipEventCreateWithFlags(&used[k], hipEventDisableTiming));
                SN_HIP_CHECK(hipEventCreateWithFlags(&used_s[k], hipEventDisableTiming));
                SN_HIP_CHECK(hipEventCreateWithFlags(&ready_r[k], hipEventDisableTiming));
                SN_HIP_CHECK(hipEventCreateWithFlags(&used_r[k], hipEventDisableTiming));
            }
            SN_HIP_CHECK(hipEventCreateWithFlags(&column, hipEventDisableTiming));
            {   // a stream created with a CU mask gets a hardware queue of its own (the runtime multiplexes the plain
                // streams of one priority onto four queues, whoever created them -- after a QZ run in the same process
                // stage 1 took 2.24 s instead of 1.64 s at n = 8000 with a plain stream here)
                hipDeviceProp_t prop; int dev = 0;
                SN_HIP_CHECK(hipGetDevice(&dev)); SN_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
                int const words = (prop.multiProcessorCount + 31) / 32;
                std::vector<uint32_t> mask(words, 0xffffffffu);
                SN_HIP_CHECK(hipExtStreamCreateWithCUMask(&pstream, words, mask.data()));
            }
            for (int k = 0; k < MAXSLOT; k++) {
                SN_HIP_CHECK(hipEventCreateWithFlags(&through[k], hipEventDisableTiming));
                SN_HIP_CHECK(hipEventCreateWithFlags(&applied[k], hipEventDisableTiming));
            }
            SN_HIP_CHECK(hipEventCreateWithFlags(&tail, hipEventDisableTiming));
            attr = true;
        }
        if (n_ <= n) return;
        release();
        n = n_;
        auto alloc = [](double *&p, size_t count) { SN_HIP_CHECK(hipMalloc((void **)&p, count * sizeof(double))); };
        alloc(V, (size_t)2 * RING * 2 * R2 * R2); alloc(VT, (size_t)2 * RING * 2 * R2 * R2);
        size_t const refl = (size_t)ht2_nslot(n) * GS * ht2_tstride(n);
        alloc(HV, refl * R2); alloc(HT, refl); alloc(GV, refl * R2); alloc(GT, refl);
        alloc(Vb, (size_t)2 * ht2_tstride(n) * 2 * R2 * R2); alloc(VTb, (size_t)2 * ht2_tstride(n) * 2 * R2 * R2);
        maxcount = n / (LAG * R2 - 1) + 4;
        alloc(gstate, (size_t)maxcount * GEN_STATE);
    }
    void release()
    {
        double **all[] = {&V, &VT, &HV, &HT, &GV, &GT, &Vb, &VTb, &gstate};
        for (double **p : all) if (*p) { SN_HIP_CHECK(hipFree(*p)); *p = nullptr; }
        n = 0;
    }
};
Ht2Workspace g_ht2;

// X_z (m x ncols_z) <- (I - V T V^T)^T X_z for one or two matrices (X1 may be NULL)
void wy_left(hipStream_t s, double const *V, double const *VT, int m, int k, double *X0, int ld0, int ncols0,
    double *X1 = nullptr, int ld1 = 0, int ncols1 = 0)
{
    if (k <= 0) return;
    if (!X1) ncols1 = 0;
    int const widest = std::max(ncols0, ncols1);
    if (widest <= 0) return;
    hipLaunchKernelGGL(ht2_wy_left_kernel, dim3(divceil(widest, WY_SLAB), X1 ? 2 : 1), dim3(WY_T), WY_LEFT_LDS, s, V, VT, m, k,
        WyTargets{{X0, X1, nullptr}, {ld0, ld1, 0}, {ncols0, ncols1, 0}});
}
// X_z (nrows_z x m) <- X_z (I - V T V^T)
void wy_right(hipStream_t s, double const *V, double const *VT, int m, int k, double *X0, int ld0, int nrows0,
    double *X1 = nullptr, int ld1 = 0, int nrows1 = 0)
{
    if (k <= 0) return;
    if (!X1) nrows1 = 0;
    int const tallest = std::max(nrows0, nrows1);
    if (tallest <= 0) return;
    hipLaunchKernelGGL(ht2_wy_right_kernel, dim3(divceil(tallest, WY_SLAB), X1 ? 2 : 1), dim3(WY_T), WY_RIGHT_LDS, s, V, VT, m, k,
        WyTargets{{X0, X1, nullptr}, {ld0, ld1, 0}, {nrows0, nrows1, 0}});
}
// the same for any subset of three targets (NULL or no rows: skipped), one launch
void wy_right3(hipStream_t s, double const *V, double const *VT, int m, int k, WyTargets tg)
{
    if (k <= 0) return;
    WyTargets live{{nullptr, nullptr, nullptr}, {0, 0, 0}, {0, 0, 0}};
    int cnt = 0, tallest = 0;
    for (int z = 0; z < 3; z++)
        if (tg.X[z] && tg.extent[z] > 0) {
            live.X[cnt] = tg.X[z]; live.ld[cnt] = tg.ld[z]; live.extent[cnt] = tg.extent[z];
            tallest = std::max(tallest, tg.extent[z]); cnt++;
        }
    if (cnt == 0) return;
    hipLaunchKernelGGL(ht2_wy_right_kernel, dim3(divceil(tallest, WY_SLAB), cnt), dim3(WY_T), WY_RIGHT_LDS, s, V, VT, m, k, live);
}

} // namespace

void ht_two_stage_release_workspace() { g_ht2.release(); }

// Can the two-stage path take this n?  (The slots of the reflector store bound the groups of sweeps in flight:
// MAXSLOT = 256 covers every n that fits the memory; the caller falls back to the rotation path otherwise.)
bool ht_two_stage_fits(int n)
{
    return n >= 3 && ht2_need_slots(n) <= MAXSLOT;
}

// (A, B), B upper triangular -> Hessenberg-triangular, Q <- Q U1, Z <- Z U2 (Q, Z may be NULL).  The reduction
// of A and B runs on `s`; Q and Z -- nothing reads them before the end -- take their transformations on `sq`
// (may be `s` itself) from a ring of factor slots, beside the next factorisations.
int ht_two_stage_device(hipStream_t s, hipStream_t sq, int n, double *A, int lda, double *B, int ldb, double *Q, int ldq,
    double *Z, int ldz, hipEvent_t between)
{
    Ht2Workspace &ws = g_ht2;
    ws.ensure(n);
    int const r = R2;
    if (!sq) sq = s;
    // the wavefronts of stage 2: sweeps jlo .. jlo + count - 1 of wavefront tau_idx (position t = tau_idx - LAG j must
    // satisfy j + 1 + t r <= n - 2); a group of GS sweeps is through with the last step of its last sweep
    int const tstride = ht2_tstride(n), nslot = ht2_nslot(n), ngroups = (n - 2 + GS - 1) / GS;
    auto wavefront = [&](int tau_idx, int &jlo, int &count) {
        int const jhi = std::min(tau_idx / LAG, n - 3);
        long const num = (long)tau_idx * r - (n - 3);
        jlo = num <= 0 ? 0 : (int)((num + (LAG * r - 1) - 1) / (LAG * r - 1));
        count = jhi - jlo + 1;
        return jlo <= jhi;
    };
    auto last_wave = [&](int g) { int const jl = std::min(g * GS + GS - 1, n - 3); return LAG * jl + (n - 3 - jl) / R2; };
    // Everything that can refuse the problem depends on n alone and is checked BEFORE the first launch: the steps
    // of a wavefront must fit the scratch of their copied blocks, the groups in flight the slots of the reflector
    // store (a dry run of the wavefront loop; an error return in the middle of it would leave kernels enqueued on
    // three streams and A, B scaled)
    {
        int opened = 0, closed = 0;
        for (int tau_idx = 0;; tau_idx++) {
            int jlo, count;
            if (!wavefront(tau_idx, jlo, count)) { if (tau_idx / LAG >= n - 3) break; else continue; }
            if (count > ws.maxcount) return -1;
            opened = std::max(opened, (jlo + count - 1) / GS + 1);
            if (opened - closed > nslot) return -2;
            while (closed < ngroups && last_wave(closed) <= tau_idx) closed++;
        }
    }
    // ---- stage 1 -----------------------------------------------------------------------------------------------
    // Two streams.  `s`: per step the application of the panel's left factor (the trailing columns of A and the rows of
    // B in one launch), the RQ factorisation of the filled block of B -- with the NEXT step's panel QR factorisation
    // as a second workgroup of the same launch (ht2_factor_kernel): it touches the panel's columns only, and nothing
    // else does until the next block column --, the application of the right factor to B and A.  `sq`: Q and Z, from two
    // rings of factor slots (V, V T^T); it tells `s` that slots are free again once per EPOCH steps, not per step
    // (two events per ring, alternating): the host's calls are counted here.
    bool const side = sq != s;
    long lcount = 0, rcount = 0;
    WyJob pendq{}; int pendq_slot = 0; long pendq_count = 0; bool have_pendq = false;   // a left factor waiting for its turn on Q
    // The application of a step's factors to Q and Z is handed to `sq` when the NEXT factorisation is launched, not
    // when its own is through: it then runs beside 92 us of latency on one CU instead of beside the HBM-bound
    // applications of the chain (which took 23 + 30 us next to it, 23 + 18 without; round 6).
    // ... and ZBATCH steps share one hand-over (one event record on the chain, 7 us, instead of one a step; `sq` lags by
    // up to ZBATCH steps behind, the rings hold RING = 8: tests/test_ht_ring_epochs.py runs the protocol).
    struct PendZ { int sl; long rc; WyJob z; bool with_q; WyJob q; long qcount; };
    constexpr int ZBATCH = 3;
    std::vector<PendZ> pendz;
    auto flush_z = [&](bool all) {
        if (pendz.empty() || (!all && (int)pendz.size() < ZBATCH)) return;
        if (side) { SN_HIP_CHECK(hipEventRecord(ws.ready_r[pendz.back().sl], s)); SN_HIP_CHECK(hipStreamWaitEvent(sq, ws.ready_r[pendz.back().sl], 0)); }
        for (PendZ const &pz : pendz) {
            if (pz.with_q) {
                hipLaunchKernelGGL(ht2_wy_right2_kernel, dim3(divceil(n, WY_SLAB), 2), dim3(WY_T), WY_RIGHT_LDS, sq, pz.q, pz.z);
                if (side && pz.qcount % (RING / 2) == RING / 2 - 1) SN_HIP_CHECK(hipEventRecord(ws.used[(pz.qcount / (RING / 2)) % 2], sq));
            } else
                wy_right(sq, pz.z.V, pz.z.VT, pz.z.m, pz.z.k, pz.z.X, pz.z.ld, pz.z.nrows);
            if (side && pz.rc % (RING / 2) == RING / 2 - 1) SN_HIP_CHECK(hipEventRecord(ws.used_r[(pz.rc / (RING / 2)) % 2], sq));
        }
        pendz.clear();
    };
    constexpr int EPOCH = RING / 2;
    // consumer side: after step i of a ring; producer side: before step L of that ring
    auto epoch_record = [&](hipEvent_t *ev, long i, hipStream_t st) { if (i % EPOCH == EPOCH - 1) SN_HIP_CHECK(hipEventRecord(ev[(i / EPOCH) % 2], st)); };
    auto epoch_wait = [&](hipEvent_t *ev, long L, hipStream_t st) { if (L >= RING && L % EPOCH == 0) SN_HIP_CHECK(hipStreamWaitEvent(st, ev[(L / EPOCH) % 2], 0)); };
    for (int jc = 0; jc < n - r - 1; jc += r) {
        int const nb = std::min(r, n - jc), top = jc + r;
        std::vector<int> starts;
        for (int i = top; i < n; i += r) starts.push_back(i);
        int const K = (int)starts.size();
        struct LeftJob { int i0, i1; };
        std::vector<LeftJob> lefts;                     // bottom up; all but a lone one (K == 1) are followed by a right step
        for (int k = K - 1; k >= 1; k--) lefts.push_back({starts[k - 1], std::min(starts[k] + r, n)});
        if (K == 1 && n - top > 1) lefts.push_back({top, n});
        // the panel QR of left step q of this block column, into slot (lcount + q - done) % RING
        long const lbase = lcount;
        auto panel_job = [&](int q) {
            long const L = lbase + q;
            int const sl = (int)(L % RING);
            if (side && Q) epoch_wait(ws.used, L, s);             // `sq` is through with the slot
            return FactorJob{1, A + (size_t)jc * lda + lefts[q].i0, lda, lefts[q].i1 - lefts[q].i0, nb,
                ws.V + (size_t)sl * 2 * r * r, ws.VT + (size_t)sl * 2 * r * r};
        };
        auto flush_q = [&]() {
            if (!have_pendq) return;
            flush_z(true);                                  // (the order of the applications to Q)
            if (side) { SN_HIP_CHECK(hipEventRecord(ws.ready[pendq_slot], s)); SN_HIP_CHECK(hipStreamWaitEvent(sq, ws.ready[pendq_slot], 0)); }
            wy_right(sq, pendq.V, pendq.VT, pendq.m, pendq.k, pendq.X, pendq.ld, pendq.nrows);
            if (side) epoch_record(ws.used, pendq_count, sq);
            have_pendq = false;
        };
        auto left_step = [&](int i0, int i1) {              // (its factor is in its slot: an earlier launch of `s`)
            int const m = i1 - i0, k = nb, sl = (int)(lcount % RING);
            double *V = ws.V + (size_t)sl * 2 * r * r, *VT = ws.VT + (size_t)sl * 2 * r * r;
            wy_left(s, V, VT, m, k, A + (size_t)(jc + nb) * lda + i0, lda, n - jc - nb, B + (size_t)i0 * ldb + i0, ldb, n - i0);
            if (Q) {
                // the application to Q waits for the right step that follows (one launch for Q and Z); a left step
                // without one flushes it by itself (flush_q)
                flush_q();
                pendq = WyJob{V, VT, m, k, Q + (size_t)i0 * ldq, ldq, n};
                pendq_slot = sl; pendq_count = lcount; have_pendq = true;
            }
            lcount++;
        };
        auto right_step = [&](int i0, int i1, int mb, FactorJob next_panel) {
            // the bottom mb rows of the block B(i0:i1, i0:i1) become [0 R]
            int const m = i1 - i0, sl = (int)(rcount % RING);
            double *V = ws.V + (size_t)(RING + sl) * 2 * r * r, *VT = ws.VT + (size_t)(RING + sl) * 2 * r * r;
            flush_z(false);                                 // earlier steps' Q / Z launches: beside THIS factorisation
            if (side && Z) epoch_wait(ws.used_r, rcount, s);
            hipLaunchKernelGGL(ht2_factor_kernel, dim3(next_panel.kind ? 2 : 1), dim3(QT), PANEL_LDS_BYTES, s,
                FactorJob{2, B + (size_t)i0 * ldb + (i1 - mb), ldb, mb, m, V, VT}, next_panel);
            wy_right(s, V, VT, m, mb, A + (size_t)i0 * lda, lda, n, B + (size_t)i0 * ldb, ldb, i1 - mb);
            if (Z) {
                pendz.push_back(PendZ{sl, rcount, WyJob{V, VT, m, mb, Z + (size_t)i0 * ldz, ldz, n}, have_pendq && mb > 0, pendq, pendq_count});
                if (pendz.back().with_q) have_pendq = false;
            }
            rcount++;
        };
        FactorJob const none{0, nullptr, 0, 0, 0, nullptr, nullptr};
        if (!lefts.empty()) {
            // (the panel's columns are final: `s` is through the previous block column)
            FactorJob const first = panel_job(0);
            hipLaunchKernelGGL(ht2_factor_kernel, dim3(1), dim3(QT), PANEL_LDS_BYTES, s, first, none);
        }
        for (int q = 0; q < (int)lefts.size(); q++) {
            int const i0 = lefts[q].i0, i1 = lefts[q].i1;
            left_step(i0, i1);
            if (K == 1) break;                              // the lone left step: the right step below is the block column's last
            right_step(i0, i1, i1 - (i0 + r), q + 1 < (int)lefts.size() ? panel_job(q + 1) : none);
        }
        int const i1 = std::min(top + r, n);
        if (i1 - top > 1) right_step(top, i1, i1 - top, none);
    }
    flush_z(true);
    if (have_pendq) {       // (the last left factor of stage 1, if no right step followed it)
        if (side) { SN_HIP_CHECK(hipEventRecord(ws.ready[pendq_slot], s)); SN_HIP_CHECK(hipStreamWaitEvent(sq, ws.ready[pendq_slot], 0)); }
        wy_right(sq, pendq.V, pendq.VT, pendq.m, pendq.k, pendq.X, pendq.ld, pendq.nrows);
        have_pendq = false;
    }
    if (between) SN_HIP_CHECK(hipEventRecord(between, s));
    // ---- stage 2 -----------------------------------------------------------------------------------------------
    // The steps of a wavefront are independent; a step depends on its own sweep's previous step and on OLDER sweeps'
    // steps of the wavefronts before.  Three launches a wavefront on `s` (the schedule is stated above the loop below).
    // (Dealing the sweeps of a wavefront to 2-4 streams so that one chain's generation hides under the
    // others' applications was correct and 2.5 x slower -- every hand-over between streams is 11 us of a loop the
    // host already bounds; round 5; removed.)
    int opened = 0, closed = 0;          // groups whose slot is claimed / whose blocks are on their way to Q, Z and the top rows
    auto close_group = [&](int g) {
        int const j0 = g * GS, gsize = std::min(GS, n - 2 - j0), slot = g % nslot;
        int const tcount = (n - 3 - j0) / R2 + 1;
        int const top = j0 + 1;             // rows [0, top) of A and B: the group's opposite reflectors are still due
        if (sq != s) { SN_HIP_CHECK(hipEventRecord(ws.through[slot], s)); SN_HIP_CHECK(hipStreamWaitEvent(sq, ws.through[slot], 0)); }
        hipLaunchKernelGGL(ht2_group_wy_kernel, dim3(tcount, 2), dim3(QT), GROUP_LDS_BYTES, sq, n, j0, gsize, tstride, slot,
            ws.HV, ws.HT, ws.GV, ws.GT, ws.Vb, ws.VTb);
        for (int t = tcount - 1; t >= 0; t--) {
            int const col0 = j0 + 1 + R2 * t, k = std::min(gsize, n - 1 - col0);
            if (k <= 0) continue;
            int const m = std::min(k - 1 + R2, n - col0);
            if (Q) wy_right(sq, ws.Vb + (size_t)t * 2 * R2 * R2, ws.VTb + (size_t)t * 2 * R2 * R2, m, k,
                Q + (size_t)col0 * ldq, ldq, n);
            // the opposite reflectors: Z, and the rows of A and B the chase left to this pass (ht2_apply_right_kernel)
            wy_right3(sq, ws.Vb + (size_t)(tstride + t) * 2 * R2 * R2, ws.VTb + (size_t)(tstride + t) * 2 * R2 * R2, m, k,
                WyTargets{{Z ? Z + (size_t)col0 * ldz : nullptr, A + (size_t)col0 * lda, B + (size_t)col0 * ldb},
                          {ldz, lda, ldb}, {n, top, top}});
        }
        if (sq != s) SN_HIP_CHECK(hipEventRecord(ws.applied[slot], sq));
    };
    // Three launches a wavefront, all on `s` (round 6):
    //   M2(tau)   = { second halves of the generations of tau | left(tau) }      left(tau): the left reflectors WITHOUT
    //                                                                             the steps' own diagonal blocks
    //   near(tau) = H on the steps' own diagonal blocks, then the near parts of the right applications: the rows from
    //               p - (r - 1) on -- everything the next wavefront's reflectors are generated from
    //   M1(tau)   = { first halves of the generations of tau' | far(tau) }       tau': the next non-empty wavefront;
    //                                                                             far(tau): the rows above p - (r - 1)
    // so that the 75 us of a generation pass beside BOTH HBM-bound applications -- a wavefront was genh + max(gen, left)
    // + right, it is max(gen / 2, far) + max(gen / 2, left) + near.  What makes it legal: the first half of gen(tau')
    // reads and writes nothing that far(tau) touches (it reads A(p + r : p + 2r, p) and B(I', I'), whose last column
    // B(p - r + 1 : p + 1, p) is the older neighbour's near part), and a step's H on its own blocks A(I, I), B(I, I)
    // can wait for near(tau) because no other operation of the wavefront touches them.  scratch/ht2_overlap.py runs
    // this order in numpy (with the negative controls: fewer than r - 1 rows above the block in the near part; the own
    // blocks left with left(tau)) and checks the entry sets of the operations that share a launch for overlaps.
    // (Tried first: the HBM-bound parts on a second stream, the generation whole on the chain -- correct, and slower
    // than before: every cross-stream hand-over is 10-14 us even when the event has long been signalled, a record
    // between two launches 7 us, and a wavefront needs four of them; stage 2 at n = 8000 2.35 -> 2.62 s.)
    std::vector<int> taus;                              // the non-empty wavefronts
    for (int tau_idx = 0;; tau_idx++) {
        int jlo, count;
        if (!wavefront(tau_idx, jlo, count)) { if (tau_idx / LAG >= n - 3) break; else continue; }
        taus.push_back(tau_idx);
    }
    auto wave_of = [&](int tau_idx) { int jlo = 0, count = 0; wavefront(tau_idx, jlo, count); return Wave2{n, tau_idx, jlo, count, tstride, nslot}; };
    auto open_slots = [&](Wave2 const &w) {            // before the first launch that writes reflectors of w
        int const jhi = w.jlo + w.count - 1;
        for (; opened <= jhi / GS; opened++) {        // a slot is free again once its previous group has been applied
            if (opened >= nslot && sq != s) SN_HIP_CHECK(hipStreamWaitEvent(s, ws.applied[opened % nslot], 0));
        }
    };
    Wave2 const none{n, 0, 0, 0, tstride, nslot};
    int const nchunk = divceil(n, LEFT_CHUNK) + 1;
    auto split_of = [&](Wave2 const &w) {              // (the bytes the left applications of the wavefront move)
        double bytes = 0.0;
        for (int k = 0; k < w.count; k++) {
            int const j = w.jlo + k, t = w.tau_idx - LAG * j, p = j + 1 + t * R2;
            if (t < 0 || j > n - 3 || p > n - 2) continue;
            bytes += 16.0 * std::min(R2, n - p) * (2.0 * (n - p) - R2);
        }
        return bytes / LEFT_BYTES_PER_US >= GEN_HIDE_US ? 0 : GEN_SPLIT;
    };
    int split = GEN_SPLIT;
    if (!taus.empty()) {
        Wave2 const w0 = wave_of(taus[0]);
        open_slots(w0);
        split = split_of(w0);
        hipLaunchKernelGGL(ht2_m1_kernel, dim3(w0.count), dim3(QT), GEN_LDS_BYTES, s, w0, split, none, 1, A, lda, B, ldb, ws.HV, ws.HT, ws.GV, ws.GT, ws.gstate);
    }
    for (size_t it = 0; it < taus.size(); it++) {
        int const tau_idx = taus[it];
        Wave2 const w = wave_of(tau_idx);
        hipLaunchKernelGGL(ht2_m2_kernel, dim3(w.count + w.count * 2 * nchunk), dim3(QT), GEN_LDS_BYTES, s, w, split, nchunk, A, lda, B, ldb,
            ws.HV, ws.HT, ws.GV, ws.GT, ws.gstate);
        hipLaunchKernelGGL(ht2_near_kernel, dim3(3, w.count, 2), dim3(256), 0, s, w, A, lda, B, ldb, ws.HV, ws.HT, ws.GV, ws.GT);
        for (; closed < ngroups && last_wave(closed) <= tau_idx; closed++) close_group(closed);
        Wave2 const wn = it + 1 < taus.size() ? wave_of(taus[it + 1]) : none;
        if (wn.count > 0) open_slots(wn);
        // (the oldest sweep of the wavefront has the smallest top: its row tiles bound the grid)
        int const base = ((w.jlo / GS) * GS + 1) & ~15, ntile4 = std::max(1, divceil(divceil(n - base, 64), 4));
        split = wn.count > 0 ? split_of(wn) : GEN_SPLIT;
        hipLaunchKernelGGL(ht2_m1_kernel, dim3(wn.count + w.count * 2 * ntile4), dim3(QT), GEN_LDS_BYTES, s, wn, split, w, ntile4, A, lda, B, ldb,
            ws.HV, ws.HT, ws.GV, ws.GT, ws.gstate);
    }
    for (; closed < ngroups; closed++) close_group(closed);
    if (sq != s) {
        SN_HIP_CHECK(hipEventRecord(ws.tail, sq));
        SN_HIP_CHECK(hipStreamWaitEvent(s, ws.tail, 0));
    }
    return 0;
}

} // namespace sn
