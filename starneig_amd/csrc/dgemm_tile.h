// The fp64 MFMA GEMM tile (v_mfma_f64_16x16x4_f64) shared by dgemm_mfma.hip and the batched
// in-place window updates of schur.hip.  See dgemm_mfma.hip for the mapping notes.
#pragma once
#include "common.h"

namespace sn {

typedef double d4 __attribute__((ext_vector_type(4)));

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

template <int BM, int BN, int KT, bool TA, bool TB>
__device__ __forceinline__
void gemm_tile(int m, int n, int k, double alpha,
    double const *__restrict__ A, int lda, double const *__restrict__ B, int ldb,
    double beta, double *__restrict__ C, int ldc, int bm, int bn, bool atomic = false)
{
    using Cfg = GemmCfg<BM, BN, KT, TA, TB>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int BUF_ELEMS = Cfg::R_ELEMS + Cfg::C_ELEMS;

    int const tid = threadIdx.x;
    int const lane = tid & 63, wave = tid >> 6;
    int const wm = wave % Cfg::WAVES_M, wn = wave / Cfg::WAVES_M;
    int const r0 = bm * BM, c0 = bn * BN;

    double rreg[Cfg::R_LOADS], creg[Cfg::C_LOADS];

    auto load_tiles = [&](int k0) {
        #pragma unroll
        for (int s = 0; s < Cfg::R_LOADS; s++) {
            int e = tid + s * 256;
            int mn, kk;
            if (TA) { kk = e % KT; mn = e / KT; } else { mn = e % BM; kk = e / BM; }
            int r = r0 + mn, kg = k0 + kk;
            double v = 0.0;
            if (r < m && kg < k)
                v = TA ? A[(size_t)r * lda + kg] : A[(size_t)kg * lda + r];
            rreg[s] = v;
        }
        #pragma unroll
        for (int s = 0; s < Cfg::C_LOADS; s++) {
            int e = tid + s * 256;
            int mn, kk;
            if (!TB) { kk = e % KT; mn = e / KT; } else { mn = e % BN; kk = e / BN; }
            int c = c0 + mn, kg = k0 + kk;
            double v = 0.0;
            if (c < n && kg < k)
                v = TB ? B[(size_t)kg * ldb + c] : B[(size_t)c * ldb + kg];
            creg[s] = v;
        }
    };
    auto store_tiles = [&](int buf) {
        double *dR = smem + buf * BUF_ELEMS, *dC = dR + Cfg::R_ELEMS;
        #pragma unroll
        for (int s = 0; s < Cfg::R_LOADS; s++) {
            int e = tid + s * 256;
            int mn, kk;
            if (TA) { kk = e % KT; mn = e / KT; dR[mn * Cfg::LDR + kk] = rreg[s]; }
            else    { mn = e % BM; kk = e / BM; dR[kk * Cfg::LDR + mn] = rreg[s]; }
        }
        #pragma unroll
        for (int s = 0; s < Cfg::C_LOADS; s++) {
            int e = tid + s * 256;
            int mn, kk;
            if (!TB) { kk = e % KT; mn = e / KT; dC[mn * Cfg::LDC + kk] = creg[s]; }
            else     { mn = e % BN; kk = e / BN; dC[kk * Cfg::LDC + mn] = creg[s]; }
        }
    };

    d4 acc[Cfg::TN][Cfg::TM];
    #pragma unroll
    for (int ci = 0; ci < Cfg::TN; ci++)
        #pragma unroll
        for (int ri = 0; ri < Cfg::TM; ri++)
            acc[ci][ri] = (d4){0.0, 0.0, 0.0, 0.0};

    int const l15 = lane & 15, l4 = lane >> 4;
    int const nkt = (k + KT - 1) / KT;

    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    for (int kt = 0; kt < nkt; kt++) {
        int const buf = kt & 1;
        if (kt + 1 < nkt) load_tiles((kt + 1) * KT);

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

        if (kt + 1 < nkt) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // epilogue: lane holds C[r = .. + l15][c = .. + l4 + 4*reg].  With beta != 0 the old values of
    // one column group are loaded together before any of them is overwritten: the loads are
    // independent (one memory latency per group instead of one per element -- the stores of
    // the plain loop may alias the next load, so the compiler cannot hoist it)
    #pragma unroll
    for (int ci = 0; ci < Cfg::TN; ci++) {
        double old[Cfg::TM][4];
        if (beta != 0.0) {
            #pragma unroll
            for (int ri = 0; ri < Cfg::TM; ri++) {
                int r = r0 + wm * Cfg::WM + ri * 16 + l15;
                #pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    int c = c0 + wn * Cfg::WN + ci * 16 + l4 + 4 * reg;
                    old[ri][reg] = (r < m && c < n) ? C[(size_t)c * ldc + r] : 0.0;
                }
            }
        }
        #pragma unroll
        for (int ri = 0; ri < Cfg::TM; ri++) {
            int r = r0 + wm * Cfg::WM + ri * 16 + l15;
            #pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                int c = c0 + wn * Cfg::WN + ci * 16 + l4 + 4 * reg;
                if (r < m && c < n) {
                    double v = alpha * acc[ci][ri][reg];
                    if (atomic) { atomicAdd(&C[(size_t)c * ldc + r], v); continue; }    // split-K slice
                    if (beta != 0.0) v += beta * old[ri][reg];
                    C[(size_t)c * ldc + r] = v;
                }
            }
        }
    }
}


} // namespace sn
