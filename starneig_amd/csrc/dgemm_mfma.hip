// fp64 GEMM on the CDNA4 matrix cores (v_mfma_f64_16x16x4_f64), column-major.
//
// This one kernel family serves every GEMM-shaped row of SURVEY.md section 8a:
//   H4/H6/H8  C -= A * B^T          (k = panel width)      -> dgemm('N','T')
//   H5        W  = A^T * (V T)      (k = trailing rows)    -> dgemm('T','N')
//   H7        W  = X * (V T)        (k = trailing cols)    -> dgemm('N','N')
//   S3        X <- lQ^T X, X <- X lQ                       -> dgemm('T','N'), ('N','N')
// replacing cblas_dgemm in hessenberg/cpu.c:315,373,433,492,552 and dgemm_ in
// common/cpu.c:96,150 (cuBLAS in hessenberg/cuda.cu:152-309, common/cuda.cu:51-153).
//
// Mapping onto the MFMA (D[i][j] += sum_k A[i][k] B[k][j], lane l holds
// A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[i=(l>>4)+4*reg][j=l&15]):
// MFMA-j is the memory-contiguous ROW index r of C, MFMA-i the column index c,
// so that the 16 lanes of a quarter-wave touch 128 contiguous bytes of C.
// MFMA-B operand = "row operand" op(A)(r,k); MFMA-A operand = "col operand" op(B)(k,c).
//
// LDS tiles are a straight copy of what is contiguous in HBM:
//   mn-contiguous operand (op(A)=A or op(B)=B^T): tile[kk][mn], ld == 16 (mod 32) doubles
//   k-contiguous  operand (op(A)=A^T or op(B)=B): tile[mn][kk], ld == KT+2 ((KT+2)/2 odd)
// both make the fragment read (16 mn x 2 k per 32-lane group, ds_read_b64)
// bank-conflict free.
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
__global__ __launch_bounds__(256, 2)
void dgemm_kernel(int m, int n, int k, double alpha,
    double const *__restrict__ A, int lda, double const *__restrict__ B, int ldb,
    double beta, double *__restrict__ C, int ldc, int tiles_m)
{
    using Cfg = GemmCfg<BM, BN, KT, TA, TB>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int BUF_ELEMS = Cfg::R_ELEMS + Cfg::C_ELEMS;

    int const tid = threadIdx.x;
    int const lane = tid & 63, wave = tid >> 6;
    int const wm = wave % Cfg::WAVES_M, wn = wave / Cfg::WAVES_M;
    int const bm = blockIdx.x % tiles_m, bn = blockIdx.x / tiles_m;
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

    // epilogue: lane holds C[r = .. + l15][c = .. + l4 + 4*reg]
    #pragma unroll
    for (int ci = 0; ci < Cfg::TN; ci++) {
        #pragma unroll
        for (int ri = 0; ri < Cfg::TM; ri++) {
            int r = r0 + wm * Cfg::WM + ri * 16 + l15;
            #pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                int c = c0 + wn * Cfg::WN + ci * 16 + l4 + 4 * reg;
                if (r < m && c < n) {
                    double *p = C + (size_t)c * ldc + r;
                    double v = alpha * acc[ci][ri][reg];
                    if (beta != 0.0) v += beta * (*p);
                    *p = v;
                }
            }
        }
    }
}

template <int BM, int BN, int KT, bool TA, bool TB>
static void launch(hipStream_t s, int m, int n, int k, double alpha,
    double const *A, int lda, double const *B, int ldb, double beta,
    double *C, int ldc)
{
    using Cfg = GemmCfg<BM, BN, KT, TA, TB>;
    static bool attr_set = false;
    auto kern = dgemm_kernel<BM, BN, KT, TA, TB>;
    if (!attr_set) {
        SN_HIP_CHECK(hipFuncSetAttribute((const void *)kern,
            hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        attr_set = true;
    }
    int tiles_m = divceil(m, BM), tiles_n = divceil(n, BN);
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), Cfg::LDS_BYTES, s,
        m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tiles_m);
}

template <bool TA, bool TB>
static void dispatch(hipStream_t s, int m, int n, int k, double alpha,
    double const *A, int lda, double const *B, int ldb, double beta,
    double *C, int ldc)
{
    // Narrow outputs (n = panel width, 280..312) waste less with 64-wide tiles;
    // few big tiles would also leave most of the 256 CUs idle.
    long tiles128 = (long)divceil(m, 128) * divceil(n, 128);
    if (tiles128 >= 512 && n >= 512)
        launch<128, 128, 16, TA, TB>(s, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
    else if ((long)divceil(m, 128) * divceil(n, 64) >= 256)
        launch<128, 64, 16, TA, TB>(s, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
    else
        launch<64, 64, 16, TA, TB>(s, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
}

// In-place window updates of the Schur path (row S3): X(w x ncols) <- U^T X and
// X(nrows x w) <- X U with a small orthogonal U (w <= 128).  One workgroup owns ALL w rows
// (resp. columns) of its output tile and reads its whole operand panel before the
// epilogue writes, so the update is safe in place (no scratch copy, unlike the reference's
// two scratch buffers per task, common/tasks.c:459-462).
void dgemm_left_inplace(hipStream_t s, int w, int ncols, double const *U, int ldu,
    double *X, int ldx)
{
    if (w <= 0 || ncols <= 0) return;
    if (w > 128) { fprintf(stderr, "[starneig-amd] dgemm_left_inplace: w > 128\n"); abort(); }
    launch<128, 128, 16, true, false>(s, w, ncols, w, 1.0, U, ldu, X, ldx, 0.0, X, ldx);
}

void dgemm_right_inplace(hipStream_t s, int nrows, int w, double const *U, int ldu,
    double *X, int ldx)
{
    if (w <= 0 || nrows <= 0) return;
    if (w > 128) { fprintf(stderr, "[starneig-amd] dgemm_right_inplace: w > 128\n"); abort(); }
    launch<128, 128, 16, false, false>(s, nrows, w, w, 1.0, X, ldx, U, ldu, 0.0, X, ldx);
}

void dgemm(hipStream_t s, char transA, char transB, int m, int n, int k,
    double alpha, double const *A, int lda, double const *B, int ldb,
    double beta, double *C, int ldc)
{
    if (m <= 0 || n <= 0) return;
    bool ta = (transA == 'T' || transA == 't');
    bool tb = (transB == 'T' || transB == 't');
    if (ta && tb)        dispatch<true, true>(s, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
    else if (ta && !tb)  dispatch<true, false>(s, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
    else if (!ta && tb)  dispatch<false, true>(s, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
    else                 dispatch<false, false>(s, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
}

} // namespace sn
