// Aggregated lazy updates of the multi-shift sweeps (row S3 of SURVEY 8a; the reference's answer to
// the same problem is the window-wide lQ that perform_push_bulges accumulates from its 50x50
// sub-windows, schur/cpu_utils.c:1920-2116, applied by common/cpu.c:54-162).
//
// A window step of a chain produces a 96 x 96 orthogonal factor U.  Applied one by one, the factors
// move 16 bytes of Q per 12 flops: the update kernels sit on the ridge of the roofline, bound by
// HBM and by the matrix cores at once.  But the LAZY zones (Q; the rows of H above the band of
// chains; the columns of the deflated part) are read by nobody until they are flushed, so only the
// ORDER of non-commuting factors matters there -- and the factors of one sweep have a regular
// structure: factor (t, c) (step t, chain c) acts on the columns of position p = t - gap c, i.e.
// [ilo + p adv, + ws), and two factors fail to commute only when |p - p'| <= r = (ws - 1) / adv.
// In the skewed coordinates u = p + r c the dependences (earlier factor -> later factor on
// overlapping columns) are componentwise non-negative in (u, c), so RECTANGLES in (u, c) are legal
// tiles: the ordered product G of the Lu x Lc factors of a tile is one orthogonal matrix of width
// adv (Lu + r (Lc - 1) - 1) + ws (396 for 4 x 4 tiles of the standard geometry: about the flops
// of the sixteen 96-wide factors, a quarter of their bytes), tiles are applied wavefront by
// wavefront (ub + cb = const: mutually independent, disjoint columns), and a tile update
//     X(:, R) <- X(:, R) G        (Q, lazy rows of H)        X(R, :) <- G^T X(R, :)   (lazy columns)
// is a k = 396 product, far above the ridge.
#pragma once
#include "common.h"
#include "schur_common.h"
#include "dgemm_tile.h"
#include <vector>
#include <algorithm>

namespace sn {

constexpr int AGG_W = 448;          // widest aggregated factor; leading dimension of every G
constexpr int AGG_LU = 5, AGG_LC = 4;        // 5 x 4 factors: 446 columns for the standard geometry
constexpr int AGG_MAXF = AGG_LU * AGG_LC;
constexpr int AGG_BM = 64;          // rows (right updates) / columns (left updates) of X per workgroup
constexpr int AGG_KT = 16;
constexpr int AGG_LDR = AGG_BM + 16, AGG_LDC = AGG_KT + 2;
constexpr int AGG_R_ELEMS = AGG_KT * AGG_LDR;     // right kernel: X tile [kk][row]
constexpr int AGG_X_ELEMS = AGG_BM * AGG_LDC;     // left kernel: X tile [col][kk]
constexpr int AGG_SLAB = 32;        // rows of G one workgroup of the build kernel owns
constexpr int AGG_BUILD_LDS = GemmCfg<AGG_SLAB, 96, 16, false, false>::LDS_BYTES;

struct AggFactor {
    double const *U;                // n x n, leading dimension 96
    int off, n;                     // columns [col0 + off, + n) of the tile
    int rs, pad;                    // lazy rows of H of the factor's own step: [0, rs), rs >= tile.rs
};
struct AggTile {
    double *G;                      // AGG_W x AGG_W scratch (ld AGG_W): the tile's ordered product
    int col0, W;                    // columns [col0, col0 + W) of Q / H
    int nfac;
    int rs;                         // lazy rows of H for this tile: [0, rs)
    AggFactor f[AGG_MAXF];
};

// conflict radius in positions and the largest tile width of a sweep geometry
__host__ __device__ inline int agg_radius(SweepStep const &st) { return (st.ws - 1) / st.adv; }
inline int agg_max_width(SweepStep const &st)
{
    return st.adv * (AGG_LU + agg_radius(st) * (AGG_LC - 1) - 1) + st.ws;
}
inline bool agg_geometry_ok(SweepStep const &st)
{
    return st.adv > 0 && st.ws <= 96 && st.gap > agg_radius(st) && agg_max_width(st) <= AGG_W;
}

// ---- G <- I, then G(:, off : off + n) <- G(:, off : off + n) U for the tile's factors in order.
// One workgroup per 32-row slab of G (the rows of G are independent).
__global__ __launch_bounds__(256)
void agg_build_kernel(AggTile const *__restrict__ tiles)
{
    AggTile const &T = tiles[blockIdx.y];
    int const W = T.W, Wp = (W + 15) & ~15;
    int const r0 = blockIdx.x * AGG_SLAB;
    if (r0 >= Wp) return;
    double *G = T.G;
    // identity in W x W, zeros in the padding up to the next multiple of 16 (the apply kernels
    // read whole 16-deep k-tiles of G)
    for (int idx = threadIdx.x; idx < AGG_SLAB * Wp; idx += 256) {
        int const r = r0 + idx % AGG_SLAB, c = idx / AGG_SLAB;
        if (r < Wp) G[(size_t)c * AGG_W + r] = (r == c && r < W) ? 1.0 : 0.0;
    }
    __syncthreads();
    int const rows = min(AGG_SLAB, W - r0);
    if (rows <= 0) return;
    int const nf = T.nfac;
    for (int i = 0; i < nf; i++) {
        AggFactor const f = T.f[i];
        double *X = G + (size_t)f.off * AGG_W + r0;
        gemm_tile<AGG_SLAB, 96, 16, false, false>(rows, f.n, f.n, 1.0, X, AGG_W, f.U, 96, 0.0, X, AGG_W, 0, 0);
        __syncthreads();            // the next factor reads what this one wrote (same workgroup, same CU)
    }
}

// ---- X(rows, R) <- X(rows, R) G, in place: a workgroup (512 threads) owns 64 rows and ALL columns of R;
// its eight waves (2 row halves x 4 column quarters, NT 16-column tiles each) keep the whole 64 x W
// result in accumulators and store it after the last k-tile has been read.  Two waves per SIMD: one
// issues MFMAs while the other waits for its LDS fragments or at the barrier.
// grid (row blocks, tiles of the wavefront); NT >= ceil(W / 64) for every tile of the launch.
// use_rs: update rows [0, tile.rs) (lazy rows of H) instead of [0, rows).
template <int NT>
__global__ __launch_bounds__(512, 2)
void agg_right_kernel(AggTile const *__restrict__ tiles, double *__restrict__ Xbase, int ldx, int rows, int use_rs)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    AggTile const &T = tiles[blockIdx.y];
    int const nrows = use_rs ? T.rs : rows;
    int const r0 = blockIdx.x * AGG_BM;
    if (r0 >= nrows) return;
    int const W = T.W;
    int const nkt = (W + 15) >> 4;              // k-tiles (G is W x W, zero-padded to a multiple of 16)
    double const *__restrict__ G = T.G;
    double *__restrict__ X = Xbase + (size_t)T.col0 * ldx;

    int const tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int const wm = wave & 1, wn = wave >> 1;
    int const l15 = lane & 15, l4 = lane >> 4;
    constexpr int C_ELEMS = 64 * NT * AGG_LDC;
    constexpr int BUF = AGG_R_ELEMS + C_ELEMS;
    bool const rows_full = r0 + AGG_BM <= nrows;

    d2 xr;                                      // X tile: 64 rows x 16 k, one row pair per thread
    d2 gr[NT];                                  // G tile: 64 NT cols x 16 k, NT k-pairs per thread

    auto load_tiles = [&](int k0) {
        {
            int const mn = (tid & 31) * 2, kk = tid >> 5;
            int const k = k0 + kk, r = r0 + mn;
            double a = 0.0, b = 0.0;
            if (k < W) {
                double const *p = X + (size_t)k * ldx + r;
                if (rows_full) { d2u v = *reinterpret_cast<d2u const *>(p); a = v.x; b = v.y; }
                else { if (r < nrows) a = p[0]; if (r + 1 < nrows) b = p[1]; }
            }
            xr = (d2){a, b};
        }
        #pragma unroll
        for (int s = 0; s < NT; s++) {
            int const e = tid + s * 512, kk = (e & 7) * 2, c = e >> 3;     // c in [64 s, 64 s + 64)
            gr[s] = *reinterpret_cast<d2 const *>(G + (size_t)c * AGG_W + k0 + kk);
        }
    };
    auto store_tiles = [&](int buf) {
        double *dR = smem + buf * BUF, *dC = dR + AGG_R_ELEMS;
        {
            int const mn = (tid & 31) * 2, kk = tid >> 5;
            *reinterpret_cast<d2 *>(dR + kk * AGG_LDR + mn) = xr;
        }
        #pragma unroll
        for (int s = 0; s < NT; s++) {
            int const e = tid + s * 512, kk = (e & 7) * 2, c = e >> 3;
            *reinterpret_cast<d2 *>(dC + c * AGG_LDC + kk) = gr[s];
        }
    };

    d4 acc[NT][2];
    #pragma unroll
    for (int ci = 0; ci < NT; ci++) { acc[ci][0] = (d4){0.0, 0.0, 0.0, 0.0}; acc[ci][1] = (d4){0.0, 0.0, 0.0, 0.0}; }

    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; kt++) {
        int const buf = kt & 1;
        if (kt + 1 < nkt) load_tiles((kt + 1) * AGG_KT);
        double const *pR = smem + buf * BUF, *pC = pR + AGG_R_ELEMS;
        #pragma unroll
        for (int ks = 0; ks < AGG_KT; ks += 4) {
            double const fr0 = pR[(ks + l4) * AGG_LDR + wm * 32 + l15];
            double const fr1 = pR[(ks + l4) * AGG_LDR + wm * 32 + 16 + l15];
            double fc[NT];
            #pragma unroll
            for (int ci = 0; ci < NT; ci++) fc[ci] = pC[((wn * NT + ci) * 16 + l15) * AGG_LDC + ks + l4];
            #pragma unroll
            for (int ci = 0; ci < NT; ci++) {
                acc[ci][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fc[ci], fr0, acc[ci][0], 0, 0, 0);
                acc[ci][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fc[ci], fr1, acc[ci][1], 0, 0, 0);
            }
        }
        if (kt + 1 < nkt) store_tiles(buf ^ 1);
        __syncthreads();
    }
    // every read of X(rows of this workgroup, R) is behind us: store the result over it
    #pragma unroll
    for (int ci = 0; ci < NT; ci++) {
        #pragma unroll
        for (int ri = 0; ri < 2; ri++) {
            int const r = r0 + wm * 32 + ri * 16 + l15;
            #pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                int const c = (wn * NT + ci) * 16 + l4 + 4 * reg;
                if (r < nrows && c < W) X[(size_t)c * ldx + r] = acc[ci][ri][reg];
            }
        }
    }
}

// ---- X(R, cols) <- G^T X(R, cols), in place: a workgroup owns 64 columns [c_lo + 64 bx, ...) and ALL
// rows of R; waves = 4 row quarters (of R, NT 16-row tiles each) x 2 column halves.  grid (column blocks, tiles).
template <int NT>
__global__ __launch_bounds__(512, 2)
void agg_left_kernel(AggTile const *__restrict__ tiles, double *__restrict__ Xbase, int ldx, int c_lo, int c_hi)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    AggTile const &T = tiles[blockIdx.y];
    int const c0 = c_lo + blockIdx.x * AGG_BM;
    if (c0 >= c_hi) return;
    int const ncols = min(AGG_BM, c_hi - c0);
    int const W = T.W;
    int const nkt = (W + 15) >> 4;
    double const *__restrict__ G = T.G;
    double *__restrict__ X = Xbase + (size_t)c0 * ldx + T.col0;    // X(k, c) = X[c * ldx + k]

    int const tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int const wm = wave >> 1, wn = wave & 1;                        // wm: row quarter of R, wn: column half
    int const l15 = lane & 15, l4 = lane >> 4;
    constexpr int G_ELEMS = 64 * NT * AGG_LDC;
    constexpr int BUF = AGG_X_ELEMS + G_ELEMS;

    d2 xr;                                      // X tile: 64 cols x 16 k, one k-pair per thread
    d2 gr[NT];                                  // G tile [i][kk] = G(k0 + kk, i)

    auto load_tiles = [&](int k0) {
        {
            int const kk = (tid & 7) * 2, c = tid >> 3;
            int const k = k0 + kk;
            double a = 0.0, b = 0.0;
            if (c < ncols) {
                double const *p = X + (size_t)c * ldx + k;
                if (k + 1 < W) { d2u v = *reinterpret_cast<d2u const *>(p); a = v.x; b = v.y; }
                else if (k < W) a = p[0];
            }
            xr = (d2){a, b};
        }
        #pragma unroll
        for (int s = 0; s < NT; s++) {
            int const e = tid + s * 512, kk = (e & 7) * 2, i = e >> 3;
            gr[s] = *reinterpret_cast<d2 const *>(G + (size_t)i * AGG_W + k0 + kk);
        }
    };
    auto store_tiles = [&](int buf) {
        double *dX = smem + buf * BUF, *dG = dX + AGG_X_ELEMS;
        {
            int const kk = (tid & 7) * 2, c = tid >> 3;
            *reinterpret_cast<d2 *>(dX + c * AGG_LDC + kk) = xr;
        }
        #pragma unroll
        for (int s = 0; s < NT; s++) {
            int const e = tid + s * 512, kk = (e & 7) * 2, i = e >> 3;
            *reinterpret_cast<d2 *>(dG + i * AGG_LDC + kk) = gr[s];
        }
    };

    d4 acc[2][NT];
    #pragma unroll
    for (int ri = 0; ri < NT; ri++) { acc[0][ri] = (d4){0.0, 0.0, 0.0, 0.0}; acc[1][ri] = (d4){0.0, 0.0, 0.0, 0.0}; }

    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; kt++) {
        int const buf = kt & 1;
        if (kt + 1 < nkt) load_tiles((kt + 1) * AGG_KT);
        double const *pX = smem + buf * BUF, *pG = pX + AGG_X_ELEMS;
        #pragma unroll
        for (int ks = 0; ks < AGG_KT; ks += 4) {
            double const fc0 = pX[(wn * 32 + l15) * AGG_LDC + ks + l4];
            double const fc1 = pX[(wn * 32 + 16 + l15) * AGG_LDC + ks + l4];
            double fr[NT];
            #pragma unroll
            for (int ri = 0; ri < NT; ri++) fr[ri] = pG[((wm * NT + ri) * 16 + l15) * AGG_LDC + ks + l4];
            #pragma unroll
            for (int ri = 0; ri < NT; ri++) {
                acc[0][ri] = __builtin_amdgcn_mfma_f64_16x16x4f64(fc0, fr[ri], acc[0][ri], 0, 0, 0);
                acc[1][ri] = __builtin_amdgcn_mfma_f64_16x16x4f64(fc1, fr[ri], acc[1][ri], 0, 0, 0);
            }
        }
        if (kt + 1 < nkt) store_tiles(buf ^ 1);
        __syncthreads();
    }
    #pragma unroll
    for (int ri = 0; ri < NT; ri++) {
        #pragma unroll
        for (int ci = 0; ci < 2; ci++) {
            int const r = (wm * NT + ri) * 16 + l15;
            #pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                int const c = wn * 32 + ci * 16 + l4 + 4 * reg;
                if (r < W && c < ncols) X[(size_t)c * ldx + r] = acc[ci][ri][reg];
            }
        }
    }
}

// launchers: the narrowest instantiation whose 64 NT columns hold the widest tile of the launch
template <int NT> constexpr int agg_right_lds() { return 2 * (AGG_R_ELEMS + 64 * NT * AGG_LDC) * 8; }
template <int NT> constexpr int agg_left_lds() { return 2 * (AGG_X_ELEMS + 64 * NT * AGG_LDC) * 8; }
inline void agg_set_attributes()
{
    SN_HIP_CHECK(hipFuncSetAttribute((const void *)agg_right_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, agg_right_lds<4>()));
    SN_HIP_CHECK(hipFuncSetAttribute((const void *)agg_right_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, agg_right_lds<5>()));
    SN_HIP_CHECK(hipFuncSetAttribute((const void *)agg_right_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, agg_right_lds<6>()));
    SN_HIP_CHECK(hipFuncSetAttribute((const void *)agg_right_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, agg_right_lds<7>()));
    SN_HIP_CHECK(hipFuncSetAttribute((const void *)agg_left_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, agg_left_lds<4>()));
    SN_HIP_CHECK(hipFuncSetAttribute((const void *)agg_left_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, agg_left_lds<5>()));
    SN_HIP_CHECK(hipFuncSetAttribute((const void *)agg_left_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, agg_left_lds<6>()));
    SN_HIP_CHECK(hipFuncSetAttribute((const void *)agg_left_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, agg_left_lds<7>()));
}
inline void agg_launch_right(hipStream_t s, int maxW, int row_blocks, int ntiles, AggTile const *tiles,
    double *X, int ldx, int rows, int use_rs)
{
    dim3 const grid(row_blocks, ntiles), block(512);
    if (maxW <= 256) hipLaunchKernelGGL(agg_right_kernel<4>, grid, block, agg_right_lds<4>(), s, tiles, X, ldx, rows, use_rs);
    else if (maxW <= 320) hipLaunchKernelGGL(agg_right_kernel<5>, grid, block, agg_right_lds<5>(), s, tiles, X, ldx, rows, use_rs);
    else if (maxW <= 384) hipLaunchKernelGGL(agg_right_kernel<6>, grid, block, agg_right_lds<6>(), s, tiles, X, ldx, rows, use_rs);
    else hipLaunchKernelGGL(agg_right_kernel<7>, grid, block, agg_right_lds<7>(), s, tiles, X, ldx, rows, use_rs);
}
inline void agg_launch_left(hipStream_t s, int maxW, int col_blocks, int ntiles, AggTile const *tiles,
    double *X, int ldx, int c_lo, int c_hi)
{
    dim3 const grid(col_blocks, ntiles), block(512);
    if (maxW <= 256) hipLaunchKernelGGL(agg_left_kernel<4>, grid, block, agg_left_lds<4>(), s, tiles, X, ldx, c_lo, c_hi);
    else if (maxW <= 320) hipLaunchKernelGGL(agg_left_kernel<5>, grid, block, agg_left_lds<5>(), s, tiles, X, ldx, c_lo, c_hi);
    else if (maxW <= 384) hipLaunchKernelGGL(agg_left_kernel<6>, grid, block, agg_left_lds<6>(), s, tiles, X, ldx, c_lo, c_hi);
    else hipLaunchKernelGGL(agg_left_kernel<7>, grid, block, agg_left_lds<7>(), s, tiles, X, ldx, c_lo, c_hi);
}

// ---- the rows of H that are lazy for a factor but not for the whole tile, [tile.rs, factor.rs): the
// factors of the tile one by one, in order, by the workgroup that owns the row block
// (grid: (row blocks of 128, tiles of the wavefront); at most a few blocks per tile are non-empty)
constexpr int AGG_LEFTOVER_LDS = GemmCfg<128, 96, 16, false, false>::LDS_BYTES;
__global__ __launch_bounds__(256, 2)
void agg_leftover_kernel(AggTile const *__restrict__ tiles, double *__restrict__ H, int ldh)
{
    AggTile const &T = tiles[blockIdx.y];
    int const rstart = T.rs + blockIdx.x * 128;
    int const nf = T.nfac;
    for (int i = 0; i < nf; i++) {
        AggFactor const f = T.f[i];
        int const rows = min(128, f.rs - rstart);
        if (rows > 0) {
            double *X = H + (size_t)(T.col0 + f.off) * ldh + rstart;
            gemm_tile<128, 96, 16, false, false>(rows, f.n, f.n, 1.0, X, ldh, f.U, 96, 0.0, X, ldh, 0, 0);
        }
        __syncthreads();
    }
}

// ---- host side: the factors of a batch of window steps, cut into tiles and wavefronts -------------
struct AggPlan {
    std::vector<AggTile> tiles;         // wavefront by wavefront
    std::vector<int> wave_begin;        // tiles [wave_begin[w], wave_begin[w + 1]) form wavefront w
    double flops_q_per_row = 0.0;       // 2 W^2 summed over the tiles (per updated row / column)
};

// steps: the SweepStep of every window step of the batch (one sweep: same geometry), in issue order;
// ubuf(i): device address of the factors of step i (ntasks blocks of 96 x 96); rsplit(i): lazy rows
// [0, rsplit(i)) of step i (non-decreasing).  G buffers are assigned by the caller.
template <typename UbufFn, typename RsFn>
inline void agg_plan(std::vector<SweepStep> const &steps, UbufFn ubuf, RsFn rsplit, AggPlan &plan,
    int LU = AGG_LU, int LC = AGG_LC)
{
    plan.tiles.clear(); plan.wave_begin.clear(); plan.flops_q_per_row = 0.0;
    if (steps.empty()) return;
    SweepStep const &g = steps.front();
    int const r = agg_radius(g), skew = g.gap - r;           // u = t - skew c
    struct Key { int ub, cb; };
    auto floordiv = [](int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
    // first pass: tile keys present
    int ub_min = 1 << 30, ub_max = -(1 << 30), cb_max = 0;
    for (SweepStep const &st : steps)
        for (int k = 0; k < st.ntasks; k++) {
            int const c = st.cmin + k, ub = floordiv(st.t - skew * c, LU);
            ub_min = std::min(ub_min, ub); ub_max = std::max(ub_max, ub); cb_max = std::max(cb_max, c / LC);
        }
    int const nub = ub_max - ub_min + 1, ncb = cb_max + 1;
    std::vector<int> slot((size_t)nub * ncb, -1);
    std::vector<AggTile> raw;
    std::vector<Key> keys;
    for (size_t i = 0; i < steps.size(); i++) {
        SweepStep const &st = steps[i];
        for (int k = 0; k < st.ntasks; k++) {
            ChaseTask const tk = make_task(st, k);
            int const c = st.cmin + k, ub = floordiv(st.t - skew * c, LU) - ub_min, cb = c / LC;
            int &s = slot[(size_t)ub * ncb + cb];
            if (s < 0) {
                s = (int)raw.size();
                AggTile t{}; t.col0 = tk.lo; t.W = tk.n; t.nfac = 0; t.rs = rsplit(i);
                raw.push_back(t); keys.push_back(Key{ub, cb});
            }
            AggTile &t = raw[s];
            // (steps come in issue order: within a tile that is a valid order of application)
            int const end = std::max(t.col0 + t.W, tk.lo + tk.n), beg = std::min(t.col0, tk.lo);
            if (tk.lo < t.col0) for (int q = 0; q < t.nfac; q++) t.f[q].off += t.col0 - tk.lo;
            t.col0 = beg; t.W = end - beg;
            t.rs = std::min(t.rs, rsplit(i));
            t.f[t.nfac++] = AggFactor{ubuf(i) + (size_t)k * 96 * 96, tk.lo - beg, tk.n, rsplit(i), 0};
        }
    }
    // wavefronts ub + cb = const
    std::vector<int> order(raw.size());
    for (size_t i = 0; i < raw.size(); i++) order[i] = (int)i;
    std::sort(order.begin(), order.end(), [&](int a, int b) {
        int const da = keys[a].ub + keys[a].cb, db = keys[b].ub + keys[b].cb;
        return da != db ? da < db : keys[a].cb < keys[b].cb;
    });
    int last = -(1 << 30);
    for (int idx : order) {
        int const d = keys[idx].ub + keys[idx].cb;
        if (d != last) { plan.wave_begin.push_back((int)plan.tiles.size()); last = d; }
        plan.tiles.push_back(raw[idx]);
        plan.flops_q_per_row += 2.0 * raw[idx].W * raw[idx].W;
    }
    plan.wave_begin.push_back((int)plan.tiles.size());
}

} // namespace sn
