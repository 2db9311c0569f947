// The fp64 MFMA GEMM tile (v_mfma_f64_16x16x4_f64) shared by dgemm_mfma.hip and the batched
// in-place window updates of schur.hip.  See dgemm_mfma.hip for the mapping notes.
#pragma once
#include "common.h"

namespace sn {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef d2 d2u __attribute__((aligned(8)));      // 16-byte global loads from 8-byte aligned addresses

template <int BM, int BN, int KT, bool TA, bool TB>
struct GemmCfg {
    static constexpr int THREADS = 256;
    static constexpr int WAVES_M = 2, WAVES_N = 2;
    static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;   // wave tile
    static constexpr int TM = WM / 16, TN = WN / 16;             // MFMA tiles per wave
    // row operand (BM x KT): k-contiguous if TA
    static constexpr int LDR = TA ? (KT + 2) : (BM + 16);
    static constexpr int R_ELEMS = TA ? BM * LDR : KT * LDR;
    // col operand (KT x BN): k-contiguous if !TB
    static constexpr int LDC = TB ? (BN + 16) : (KT + 2);
    static constexpr int C_ELEMS = TB ? KT * LDC : BN * LDC;
    static constexpr int R_LOADS = BM * KT / THREADS;
    static constexpr int C_LOADS = BN * KT / THREADS;
    static constexpr int LDS_BYTES = 2 * (R_ELEMS + C_ELEMS) * 8;
};

// FLUSH > 0: two-level summation -- the accumulators are folded into a second set every FLUSH
// k-tiles (FLUSH * KT terms) and restart from zero, so a sum over a long k is a short chain of
// chunk sums instead of one chain of k / 4 roundings at the magnitude of the result (the rounding
// error of a length-k chain grows like sqrt(k); measured on the accumulation of Q: DESIGN.md).
template <int BM, int BN, int KT, bool TA, bool TB, int FLUSH = 0>
__device__ __forceinline__
void gemm_tile(int m, int n, int k, double alpha,
    double const *__restrict__ A, int lda, double const *__restrict__ B, int ldb,
    double beta, double *__restrict__ C, int ldc, int bm, int bn, bool separate_sum = false)
{
    using Cfg = GemmCfg<BM, BN, KT, TA, TB>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int BUF_ELEMS = Cfg::R_ELEMS + Cfg::C_ELEMS;

    int const tid = threadIdx.x;
    int const lane = tid & 63, wave = tid >> 6;
    int const wm = wave % Cfg::WAVES_M, wn = wave / Cfg::WAVES_M;
    int const r0 = bm * BM, c0 = bn * BN;

    double rreg[Cfg::R_LOADS], creg[Cfg::C_LOADS];

    // C <- C + alpha op(A) op(B) with alpha = +-1 (the rank-k updates of the Hessenberg path): the
    // accumulators START as the old C tile and the column operand carries the sign, so the tile
    // is read while the first operand tiles are in flight and the epilogue is stores only --
    // instead of a dependent load / fma / store tail after the last MFMA.
    bool const accinit = !separate_sum && beta == 1.0 && (alpha == 1.0 || alpha == -1.0);
    double const opscale = accinit ? alpha : 1.0;
    // interior tiles: straight 16-byte loads, no per-element bounds tests
    bool const tile_full = (r0 + BM <= m) && (c0 + BN <= n);

    // Boundary tiles / the partial last k-tile: the loads are UNCONDITIONAL from clamped addresses
    // (always inside the operands) and the out-of-range entries are replaced by zeros where the
    // tile goes to LDS.  (Predicated loads -- `v = 0; if (in range) v = load` -- made the compiler
    // wait for every earlier load before each zero-initialisation: sixteen memory latencies per
    // k-tile in a row, and the boundary tiles ended the launch late.)
    auto load_tiles = [&](int k0) {
        if (tile_full && k0 + KT <= k) {
            #pragma unroll
            for (int s = 0; s < Cfg::R_LOADS / 2; s++) {
                int e = tid + s * 256;
                d2u v;
                if (TA) { int kk = (e % (KT / 2)) * 2, mn = e / (KT / 2);
                          v = *reinterpret_cast<d2u const *>(A + (size_t)(r0 + mn) * lda + k0 + kk); }
                else    { int mn = (e % (BM / 2)) * 2, kk = e / (BM / 2);
                          v = *reinterpret_cast<d2u const *>(A + (size_t)(k0 + kk) * lda + r0 + mn); }
                rreg[2 * s] = v.x; rreg[2 * s + 1] = v.y;
            }
            #pragma unroll
            for (int s = 0; s < Cfg::C_LOADS / 2; s++) {
                int e = tid + s * 256;
                d2u v;
                if (!TB) { int kk = (e % (KT / 2)) * 2, mn = e / (KT / 2);
                           v = *reinterpret_cast<d2u const *>(B + (size_t)(c0 + mn) * ldb + k0 + kk); }
                else     { int mn = (e % (BN / 2)) * 2, kk = e / (BN / 2);
                           v = *reinterpret_cast<d2u const *>(B + (size_t)(k0 + kk) * ldb + c0 + mn); }
                creg[2 * s] = v.x; creg[2 * s + 1] = v.y;
            }
            return;
        }
        #pragma unroll
        for (int s = 0; s < Cfg::R_LOADS / 2; s++) {
            #pragma unroll
            for (int h = 0; h < 2; h++) {
                int e = tid + s * 256;
                int mn, kk;
                if (TA) { kk = (e % (KT / 2)) * 2 + h; mn = e / (KT / 2); }
                else    { mn = (e % (BM / 2)) * 2 + h; kk = e / (BM / 2); }
                int r = min(r0 + mn, m - 1), kg = min(k0 + kk, k - 1);
                rreg[2 * s + h] = TA ? A[(size_t)r * lda + kg] : A[(size_t)kg * lda + r];
            }
        }
        #pragma unroll
        for (int s = 0; s < Cfg::C_LOADS / 2; s++) {
            #pragma unroll
            for (int h = 0; h < 2; h++) {
                int e = tid + s * 256;
                int mn, kk;
                if (!TB) { kk = (e % (KT / 2)) * 2 + h; mn = e / (KT / 2); }
                else     { mn = (e % (BN / 2)) * 2 + h; kk = e / (BN / 2); }
                int c = min(c0 + mn, n - 1), kg = min(k0 + kk, k - 1);
                creg[2 * s + h] = TB ? B[(size_t)kg * ldb + c] : B[(size_t)c * ldb + kg];
            }
        }
    };
    // k0: the k-offset of the tile held in rreg / creg
    auto store_tiles = [&](int buf, int k0) {
        double *dR = smem + buf * BUF_ELEMS, *dC = dR + Cfg::R_ELEMS;
        bool const full = tile_full && k0 + KT <= k;
        #pragma unroll
        for (int s = 0; s < Cfg::R_LOADS / 2; s++) {
            int e = tid + s * 256;
            double *d;
            int mn, kk;
            if (TA) { kk = (e % (KT / 2)) * 2; mn = e / (KT / 2); d = dR + mn * Cfg::LDR + kk; }
            else    { mn = (e % (BM / 2)) * 2; kk = e / (BM / 2); d = dR + kk * Cfg::LDR + mn; }
            double x = rreg[2 * s], y = rreg[2 * s + 1];
            if (!full) {
                bool const in0 = (r0 + mn < m) && (k0 + kk < k);
                bool const in1 = TA ? ((r0 + mn < m) && (k0 + kk + 1 < k)) : ((r0 + mn + 1 < m) && (k0 + kk < k));
                x = in0 ? x : 0.0; y = in1 ? y : 0.0;
            }
            *reinterpret_cast<d2 *>(d) = (d2){x, y};
        }
        #pragma unroll
        for (int s = 0; s < Cfg::C_LOADS / 2; s++) {
            int e = tid + s * 256;
            double *d;
            int mn, kk;
            if (!TB) { kk = (e % (KT / 2)) * 2; mn = e / (KT / 2); d = dC + mn * Cfg::LDC + kk; }
            else     { mn = (e % (BN / 2)) * 2; kk = e / (BN / 2); d = dC + kk * Cfg::LDC + mn; }
            // (the sign of the column operand is applied HERE, not where the tile is loaded: a use of
            // the loaded registers before the MFMA block makes the compiler wait for the global loads
            // of the next k-tile before it issues this k-tile's MFMAs)
            double x = creg[2 * s] * opscale, y = creg[2 * s + 1] * opscale;
            if (!full) {
                bool const in0 = (c0 + mn < n) && (k0 + kk < k);
                bool const in1 = !TB ? ((c0 + mn < n) && (k0 + kk + 1 < k)) : ((c0 + mn + 1 < n) && (k0 + kk < k));
                x = in0 ? x : 0.0; y = in1 ? y : 0.0;
            }
            *reinterpret_cast<d2 *>(d) = (d2){x, y};
        }
    };

    d4 acc[Cfg::TN][Cfg::TM];
    constexpr bool TWO = FLUSH > 0;
    d4 acc2[TWO ? Cfg::TN : 1][TWO ? Cfg::TM : 1];
    #pragma unroll
    for (int ci = 0; ci < Cfg::TN; ci++)
        #pragma unroll
        for (int ri = 0; ri < Cfg::TM; ri++) {
            acc[ci][ri] = (d4){0.0, 0.0, 0.0, 0.0};
            if (TWO) acc2[TWO ? ci : 0][TWO ? ri : 0] = (d4){0.0, 0.0, 0.0, 0.0};
        }

    int const l15 = lane & 15, l4 = lane >> 4;
    int const nkt = (k + KT - 1) / KT;

    if (accinit) {
        #pragma unroll
        for (int ci = 0; ci < Cfg::TN; ci++)
            #pragma unroll
            for (int ri = 0; ri < Cfg::TM; ri++) {
                int r = r0 + wm * Cfg::WM + ri * 16 + l15;
                #pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    int c = c0 + wn * Cfg::WN + ci * 16 + l4 + 4 * reg;
                    if (tile_full || (r < m && c < n)) {
                        if (TWO) acc2[TWO ? ci : 0][TWO ? ri : 0][reg] = C[(size_t)c * ldc + r];
                        else acc[ci][ri][reg] = C[(size_t)c * ldc + r];
                    }
                }
            }
    }
    // Three stages with ONE register set and two LDS buffers: while the MFMAs of k-tile kt run out of
    // LDS[kt & 1], tile kt + 1 goes from the registers to the other buffer FIRST (its loads were
    // issued a whole block ago, and that buffer was released by the barrier that ended block kt - 1)
    // and the loads of tile kt + 2 are issued into the registers just freed.  The end of a block is
    // then the barrier alone -- not "wait for the loads, write LDS, wait, barrier" with the matrix
    // pipe idle meanwhile.  (PMC, 20000^2 x 624, two workgroups per CU: the waves were parked at
    // s_waitcnt / s_barrier for 13 % of their cycles and the two waves of a SIMD reach that tail
    // together: 74 % MFMA busy.)
    load_tiles(0);
    store_tiles(0, 0);
    if (nkt > 1) load_tiles(KT);
    __syncthreads();

    for (int kt = 0; kt < nkt; kt++) {
        int const buf = kt & 1;
        if (kt + 1 < nkt) store_tiles(buf ^ 1, (kt + 1) * KT);
        if (kt + 2 < nkt) load_tiles((kt + 2) * KT);

        double const *pR = smem + buf * BUF_ELEMS, *pC = pR + Cfg::R_ELEMS;
        #pragma unroll
        for (int ks = 0; ks < KT; ks += 4) {
            double fr[Cfg::TM], fc[Cfg::TN];
            #pragma unroll
            for (int ri = 0; ri < Cfg::TM; ri++) {
                int mn = wm * Cfg::WM + ri * 16 + l15;
                fr[ri] = TA ? pR[mn * Cfg::LDR + ks + l4] : pR[(ks + l4) * Cfg::LDR + mn];
            }
            #pragma unroll
            for (int ci = 0; ci < Cfg::TN; ci++) {
                int mn = wn * Cfg::WN + ci * 16 + l15;
                fc[ci] = !TB ? pC[mn * Cfg::LDC + ks + l4] : pC[(ks + l4) * Cfg::LDC + mn];
            }
            #pragma unroll
            for (int ci = 0; ci < Cfg::TN; ci++)
                #pragma unroll
                for (int ri = 0; ri < Cfg::TM; ri++)
                    acc[ci][ri] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                        fc[ci], fr[ri], acc[ci][ri], 0, 0, 0);
        }

        if (TWO && ((kt + 1) % (TWO ? FLUSH : 1) == 0 || kt + 1 == nkt)) {
            #pragma unroll
            for (int ci = 0; ci < Cfg::TN; ci++)
                #pragma unroll
                for (int ri = 0; ri < Cfg::TM; ri++) {
                    acc2[TWO ? ci : 0][TWO ? ri : 0] += acc[ci][ri];
                    acc[ci][ri] = (kt + 1 == nkt) ? acc2[TWO ? ci : 0][TWO ? ri : 0] : (d4){0.0, 0.0, 0.0, 0.0};
                }
        }
        __syncthreads();
    }

    // epilogue: lane holds C[r = .. + l15][c = .. + l4 + 4*reg].  With beta != 0 the old values of
    // one column group are loaded together before any of them is overwritten: the loads are
    // independent (one memory latency per group instead of one per element -- the stores of
    // the plain loop may alias the next load, so the compiler cannot hoist it)
    if (accinit) {
        #pragma unroll
        for (int ci = 0; ci < Cfg::TN; ci++)
            #pragma unroll
            for (int ri = 0; ri < Cfg::TM; ri++) {
                int r = r0 + wm * Cfg::WM + ri * 16 + l15;
                #pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    int c = c0 + wn * Cfg::WN + ci * 16 + l4 + 4 * reg;
                    if (tile_full || (r < m && c < n)) C[(size_t)c * ldc + r] = acc[ci][ri][reg];
                }
            }
        return;
    }
    if (beta != 0.0) {
        // C <- alpha * sum + beta * C: the old values of column group ci + 1 are requested BEFORE group ci
        // is stored (the groups are different columns), so the epilogue costs one memory latency, not
        // one per group; the loads are unconditional from clamped addresses (a predicated load makes
        // the compiler wait for every earlier one: see load_tiles)
        double old[2][Cfg::TM][4];
        auto fetch = [&](int ci, double (&o)[Cfg::TM][4]) {
            #pragma unroll
            for (int ri = 0; ri < Cfg::TM; ri++) {
                int const r = min(r0 + wm * Cfg::WM + ri * 16 + l15, m - 1);
                #pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    int const c = min(c0 + wn * Cfg::WN + ci * 16 + l4 + 4 * reg, n - 1);
                    o[ri][reg] = C[(size_t)c * ldc + r];
                }
            }
        };
        fetch(0, old[0]);
        #pragma unroll
        for (int ci = 0; ci < Cfg::TN; ci++) {
            if (ci + 1 < Cfg::TN) fetch(ci + 1, old[(ci + 1) & 1]);
            #pragma unroll
            for (int ri = 0; ri < Cfg::TM; ri++) {
                int const r = r0 + wm * Cfg::WM + ri * 16 + l15;
                #pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    int const c = c0 + wn * Cfg::WN + ci * 16 + l4 + 4 * reg;
                    if (tile_full || (r < m && c < n))
                        C[(size_t)c * ldc + r] = alpha * acc[ci][ri][reg] + beta * old[ci & 1][ri][reg];
                }
            }
        }
        return;
    }
    #pragma unroll
    for (int ci = 0; ci < Cfg::TN; ci++) {
        #pragma unroll
        for (int ri = 0; ri < Cfg::TM; ri++) {
            int r = r0 + wm * Cfg::WM + ri * 16 + l15;
            #pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                int c = c0 + wn * Cfg::WN + ci * 16 + l4 + 4 * reg;
                if (r < m && c < n) {
                    double v = alpha * acc[ci][ri][reg];
                    C[(size_t)c * ldc + r] = v;
                }
            }
        }
    }
}


} // namespace sn
