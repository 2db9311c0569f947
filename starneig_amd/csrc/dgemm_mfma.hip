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
#include "dgemm_tile.h"
#include "tuning.h"
#include <algorithm>
#include <vector>

namespace sn {

// tile id -> (bm, bn).  Tile order (speed only): blocks are dealt round-robin over the 8 XCDs, so give
// every XCD a contiguous range of tile ids (its private L2 then sees neighbouring tiles), and walk the
// tiles in groups of 8 tile-rows so that the ~64 tiles resident on one XCD cover an 8 x 8 patch: 16
// operand panels instead of 65.
__device__ __forceinline__ void tile_of(int pid, int total, int tiles_m, int &bm, int &bn)
{
    int const tiles_n = total / tiles_m;
    int const cpx = total / 8;
    if (pid < cpx * 8) pid = (pid % 8) * cpx + pid / 8;
    constexpr int GROUP_M = 8;
    int const in_group = GROUP_M * tiles_n;
    int const group = pid / in_group, first_m = group * GROUP_M;
    int const gsize = min(tiles_m - first_m, GROUP_M);
    bm = first_m + (pid % in_group) % gsize; bn = (pid % in_group) / gsize;
}

// SPLIT: the tiles of the last, partly filled round of workgroups (the chip holds 512 of them) are cut
// into `pieces` = 2 (halves, BM/2 x BN) or 4 (quarters) so that this round fills the chip too: the first
// `whole` blocks take whole tiles, the others one piece each of the tiles whole, whole + 1, ...  (the
// trailing matrix of the Hessenberg reduction shrinks panel by panel: at m = 10000 a launch is 12.2
// rounds of tiles and paid for 13)
template <int BM, int BN, int KT, bool TA, bool TB, int FLUSH, bool SPLIT>
__global__ __launch_bounds__(256, 2)
void dgemm_kernel(int m, int n, int k, double alpha,
    double const *__restrict__ A, int lda, double const *__restrict__ B, int ldb,
    double beta, double *__restrict__ C, int ldc, int tiles_m, int separate_sum, int total, int whole, int pieces)
{
    int bm, bn;
    if (!SPLIT || (int)blockIdx.x < whole) {
        tile_of(blockIdx.x, total, tiles_m, bm, bn);
        gemm_tile<BM, BN, KT, TA, TB, FLUSH>(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, bm, bn, separate_sum != 0);
        return;
    }
    if (SPLIT) {
        int const q = blockIdx.x - whole, piece = q % pieces;
        tile_of(whole + q / pieces, total, tiles_m, bm, bn);
        if (pieces == 2)
            gemm_tile<BM / 2, BN, KT, TA, TB, FLUSH>(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, 2 * bm + piece, bn, separate_sum != 0);
        else
            gemm_tile<BM / 2, BN / 2, KT, TA, TB, FLUSH>(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, 2 * bm + (piece & 1), 2 * bn + (piece >> 1), separate_sum != 0);
    }
}

// Split-K form for outputs with few tiles and a long inner dimension (the inner-product shaped
// GEMMs of the Hessenberg path: W = A^T (V T), W = X (V T), S = Y^T (V T) -- skinny n = panel
// width, k = trailing rows): blockIdx.y selects a slice of k; slice y writes ITS product into plane y of a
// scratch buffer (m x n, leading dimension m, plain stores) and dgemm_splitk_sum_kernel adds the planes up in
// slice order.  Besides filling the chip this shortens the sequential accumulation chains from k to
// k / slices terms.  (Rounds 1-5 summed the slices into C with fp64 atomics: the order of the partial sums, and
// with it the last bits of every Hessenberg reduction, varied from run to run and from replica to replica.)
template <int BM, int BN, int KT, bool TA, bool TB, int FLUSH>
__global__ __launch_bounds__(256, 2)
void dgemm_splitk_kernel(int m, int n, int k, int kchunk, double alpha,
    double const *__restrict__ A, int lda, double const *__restrict__ B, int ldb,
    double *__restrict__ planes, int tiles_m)
{
    double *C = planes + (size_t)blockIdx.y * m * n;
    int const ldc = m;
    int const k0 = blockIdx.y * kchunk, kl = min(kchunk, k - k0);
    if (kl <= 0) return;
    // Tile order (speed only): the few column tiles of ONE row panel go to the same XCD back to back
    // (blocks b and b + 8 share an XCD), so the long operand panel -- 128 columns of the trailing matrix
    // in W = At^T (V T) -- comes from HBM once and from that XCD's L2 for the other column tiles.
    // With fewer than 8 row panels that order would leave whole XCDs without work (one row panel: every active block
    // on XCD 0 -- the 64 x n products of the QR step ran on 32 CUs until round 6): there the column tiles go round the
    // XCDs instead.
    int const tiles_n = (n + BN - 1) / BN;
    int bm, bn;
    if (tiles_m >= 8) {
        int const x = blockIdx.x % 8, j = blockIdx.x / 8;
        bm = x + 8 * (j / tiles_n); bn = j % tiles_n;
        if (bm >= tiles_m) return;
    } else {
        bm = blockIdx.x % tiles_m; bn = blockIdx.x / tiles_m;
    }
    double const *Ak = TA ? A + k0 : A + (size_t)k0 * lda;
    double const *Bk = TB ? B + (size_t)k0 * ldb : B + k0;
    gemm_tile<BM, BN, KT, TA, TB, FLUSH>(m, n, kl, alpha, Ak, lda, Bk, ldb, 0.0, C, ldc, bm, bn);
}

// C(r, c) = plane_0(r, c) + plane_1(r, c) + ... in slice order: one thread per row pair, all planes in flight
__global__ __launch_bounds__(256)
void dgemm_splitk_sum_kernel(int m, int n, int slices, double const *__restrict__ planes, double *__restrict__ C, int ldc)
{
    int const r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    size_t const plane = (size_t)m * n;
    for (int c = blockIdx.y; c < n; c += gridDim.y) {
        double const *p = planes + (size_t)c * m + r;
        double s = p[0];
        int y = 1;
        for (; y + 4 <= slices; y += 4) {
            double const a = p[(size_t)y * plane], b = p[(size_t)(y + 1) * plane], d = p[(size_t)(y + 2) * plane], e = p[(size_t)(y + 3) * plane];
            s += a; s += b; s += d; s += e;
        }
        for (; y < slices; y++) s += p[(size_t)y * plane];
        C[(size_t)c * ldc + r] = s;
    }
}

// scratch planes of the split-K products: one buffer per (host thread, stream) -- the critical stream and the
// side stream of a reduction run their products concurrently
struct SplitkSlot { hipStream_t s; double *p; size_t cap; };
static thread_local std::vector<SplitkSlot> g_splitk_slots;
void dgemm_release_workspace()
{
    for (SplitkSlot &q : g_splitk_slots) if (q.p) SN_HIP_CHECK(hipFree(q.p));
    g_splitk_slots.clear();
}
static double *splitk_planes(hipStream_t s, size_t doubles)
{
    typedef SplitkSlot Slot;
    std::vector<Slot> &slots = g_splitk_slots;
    for (Slot &q : slots)
        if (q.s == s) {
            if (q.cap < doubles) {
                SN_HIP_CHECK(hipStreamSynchronize(s));
                SN_HIP_CHECK(hipFree(q.p));
                SN_HIP_CHECK(hipMalloc((void **)&q.p, doubles * sizeof(double))); q.cap = doubles;
            }
            return q.p;
        }
    Slot q{s, nullptr, doubles};
    SN_HIP_CHECK(hipMalloc((void **)&q.p, doubles * sizeof(double)));
    slots.push_back(q);
    return q.p;
}

// Batched form: blockIdx.y selects a problem descriptor (alpha = 1, beta = 0).  Used for the
// in-place window updates of all bulge chains of one sweep step in a single launch.
template <int BM, int BN, int KT, bool TA, bool TB>
__global__ __launch_bounds__(256, 2)
void dgemm_batched_kernel(GemmDesc const *__restrict__ descs)
{
    GemmDesc const d = descs[blockIdx.y];
    int const tiles_m = (d.m + BM - 1) / BM, tiles_n = (d.n + BN - 1) / BN;
    if ((int)blockIdx.x >= tiles_m * tiles_n) return;
    gemm_tile<BM, BN, KT, TA, TB>(d.m, d.n, d.k, 1.0, d.A, d.lda, d.B, d.ldb, 0.0, d.C, d.ldc,
        blockIdx.x % tiles_m, blockIdx.x / tiles_m);
}

template <int BM, int BN, int KT, bool TA, bool TB>
static void launch_batched(hipStream_t s, GemmDesc const *ddescs, int count, int max_tiles)
{
    using Cfg = GemmCfg<BM, BN, KT, TA, TB>;
    static thread_local bool attr_set = false;      // per host thread = per device
    auto kern = dgemm_batched_kernel<BM, BN, KT, TA, TB>;
    if (!attr_set) {
        SN_HIP_CHECK(hipFuncSetAttribute((const void *)kern,
            hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        attr_set = true;
    }
    if (count <= 0 || max_tiles <= 0) return;
    hipLaunchKernelGGL(kern, dim3(max_tiles, count), dim3(256), Cfg::LDS_BYTES, s, ddescs);
}

// descs[i]: C(w x ncols) <- A^T(w x w) * B(w x ncols) with C == B   (left updates), or
//           C(nrows x w) <- A(nrows x w) * B(w x w)   with C == A   (right updates);
// every problem must have its in-place dimension <= 128 (one tile).
void dgemm_batched_left_inplace(hipStream_t s, GemmDesc const *ddescs, int count, int max_cols)
{
    launch_batched<128, 128, 16, true, false>(s, ddescs, count, divceil(max_cols, 128));
}
void dgemm_batched_right_inplace(hipStream_t s, GemmDesc const *ddescs, int count, int max_rows)
{
    launch_batched<128, 128, 16, false, false>(s, ddescs, count, divceil(max_rows, 128));
}

template <int BM, int BN, int KT, bool TA, bool TB>
static void launch(hipStream_t s, int m, int n, int k, double alpha,
    double const *A, int lda, double const *B, int ldb, double beta,
    double *C, int ldc, bool split_ok = false)
{
    using Cfg = GemmCfg<BM, BN, KT, TA, TB>;
    static thread_local bool attr_set = false;      // per host thread = per device
    // two-level summation (chunks of 256 terms) wherever the second accumulator set fits the
    // register budget of two workgroups per CU: every tile shape but 128 x 128
    constexpr int FLUSH = (BM * BN <= 128 * 64) ? 256 / KT : 0;
    // the big tile cuts the tiles of its last round of workgroups into pieces (see the kernel)
    constexpr bool SPLIT = (BM == 128 && BN == 128);
    auto kern = dgemm_kernel<BM, BN, KT, TA, TB, FLUSH, SPLIT>;
    if (!attr_set) {
        SN_HIP_CHECK(hipFuncSetAttribute((const void *)kern,
            hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        attr_set = true;
    }
    int const tiles_m = divceil(m, BM), tiles_n = divceil(n, BN), total = tiles_m * tiles_n;
    int whole = total, pieces = 1;
    // (never for the in-place window updates below: there ONE workgroup must own all rows / columns of its tile)
    if (SPLIT && split_ok && !tuning().gemm_nosplit) {
        int const slots = 512, r = total % slots;
        if (total > slots && r > 0 && r <= slots / 2) { pieces = (r <= slots / 4) ? 4 : 2; whole = total - r; }
    }
    hipLaunchKernelGGL(kern, dim3(whole + (total - whole) * pieces), dim3(256), Cfg::LDS_BYTES, s,
        m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tiles_m, tuning().gemm_separate_sum ? 1 : 0, total, whole, pieces);
}

template <int BM, int BN, bool TA, bool TB>
static void launch_splitk(hipStream_t s, int m, int n, int k, int slices, double alpha,
    double const *A, int lda, double const *B, int ldb, double *C, int ldc)
{
    using Cfg = GemmCfg<BM, BN, 16, TA, TB>;
    static thread_local bool attr_set = false;      // per host thread = per device
    auto kern = dgemm_splitk_kernel<BM, BN, 16, TA, TB, 256 / 16>;
    if (!attr_set) {
        SN_HIP_CHECK(hipFuncSetAttribute((const void *)kern,
            hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        attr_set = true;
    }
    int const kchunk = (int)roundup((size_t)divceil(k, slices), 16);
    int const tiles_m = divceil(m, BM);
    int const tiles = (tiles_m >= 8 ? 8 * divceil(tiles_m, 8) : tiles_m) * divceil(n, BN);   // (rounded up: see the kernel's tile order)
    int const nsl = divceil(k, kchunk);
    double *planes = splitk_planes(s, (size_t)nsl * m * n);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles, nsl), dim3(256), Cfg::LDS_BYTES, s,
        m, n, k, kchunk, alpha, A, lda, B, ldb, planes, tiles_m);
    hipLaunchKernelGGL(dgemm_splitk_sum_kernel, dim3(divceil(m, 256), std::min(n, 256)), dim3(256), 0, s, m, n, nsl, planes, C, ldc);
}

template <bool TA, bool TB>
static void dispatch(hipStream_t s, int m, int n, int k, double alpha,
    double const *A, int lda, double const *B, int ldb, double beta,
    double *C, int ldc)
{
    // (only inner-product shapes, min(m,n) << k: the square window products of the Schur path must
    // stay deterministic -- replicas of one reduction on several GPUs rely on it)
    if (beta == 0.0 && k >= 2048 && (long)std::min(m, n) * 4 <= k) {
        // Few output tiles, long k: split k over the chip (the slices' products are added in slice order:
        // the same bits in every run, unlike the reference's STARPU_COMMUTE sums).
        // Target: ~6 work items per workgroup slot (256 CUs x 2), slices of >= 512.
        long const t128 = (long)divceil(m, 128) * divceil(n, 64);
        long const t64 = (long)divceil(m, 64) * divceil(n, 64);
        int const kchunk = tuning().gemm_kchunk;
        if (kchunk > 0 && t64 > 64) {
            launch_splitk<128, 64, TA, TB>(s, m, n, k, divceil(k, kchunk), alpha, A, lda, B, ldb, C, ldc);
            return;
        }
        if (t64 <= 64 || m <= 64) {       // (one row of 64 x 64 tiles: W = (V T)^T A of the QR step, bound by reading A)
            int const slices = (int)std::min<long>(divceil(k, 512), std::max<long>(1, (t64 <= 64 ? 1024 : 2048) / t64));
            launch_splitk<64, 64, TA, TB>(s, m, n, k, slices, alpha, A, lda, B, ldb, C, ldc);
            return;
        }
        if (t128 < 2048) {
            int const slices = (int)std::min<long>(divceil(k, 512), divceil(3072, (int)t128));
            if (slices > 1) {
                launch_splitk<128, 64, TA, TB>(s, m, n, k, slices, alpha, A, lda, B, ldb, C, ldc);
                return;
            }
        }
    }
    // Narrow outputs (n = panel width, 280..312) waste less with 64-wide tiles;
    // few big tiles would also leave most of the 256 CUs idle.
    long tiles128 = (long)divceil(m, 128) * divceil(n, 128);
    if (tiles128 >= 512 && n >= 512)
        launch<128, 128, 16, TA, TB>(s, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, A != C && B != C);
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

// k <= 0: the product is empty, C <- beta C (BLAS semantics).  The tile kernels load their boundary k-tiles
// from clamped addresses min(k0 + kk, k - 1) and must never see k = 0.
__global__ void dgemm_scale_kernel(int m, int n, double beta, double *__restrict__ C, int ldc)
{
    int const r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    for (int c = blockIdx.y; c < n; c += gridDim.y) C[(size_t)c * ldc + r] *= beta;
}
static void empty_product(hipStream_t s, int m, int n, double beta, double *C, int ldc)
{
    if (beta == 1.0) return;
    if (beta == 0.0) { SN_HIP_CHECK(hipMemset2DAsync(C, (size_t)ldc * sizeof(double), 0, (size_t)m * sizeof(double), n, s)); return; }
    hipLaunchKernelGGL(dgemm_scale_kernel, dim3(divceil(m, 256), std::min(n, 1024)), dim3(256), 0, s, m, n, beta, C, ldc);
}

// C = op(A) op(B) with 128 x 64 tiles and two-level summation whatever the shape: the acceptance
// checks (||Q Q^T - I||, ||Q H Q^T - A||) must not add rounding noise of their own -- a single
// chain over k = n = 20000 costs ~10 u on the diagonal of Q Q^T
void dgemm_accurate(hipStream_t s, char transA, char transB, int m, int n, int k,
    double const *A, int lda, double const *B, int ldb, double *C, int ldc)
{
    if (m <= 0 || n <= 0) return;
    if (k <= 0) { empty_product(s, m, n, 0.0, C, ldc); return; }
    bool ta = (transA == 'T' || transA == 't'), tb = (transB == 'T' || transB == 't');
    if (ta && tb)        launch<128, 64, 16, true, true>(s, m, n, k, 1.0, A, lda, B, ldb, 0.0, C, ldc);
    else if (ta && !tb)  launch<128, 64, 16, true, false>(s, m, n, k, 1.0, A, lda, B, ldb, 0.0, C, ldc);
    else if (!ta && tb)  launch<128, 64, 16, false, true>(s, m, n, k, 1.0, A, lda, B, ldb, 0.0, C, ldc);
    else                 launch<128, 64, 16, false, false>(s, m, n, k, 1.0, A, lda, B, ldb, 0.0, C, ldc);
}

void dgemm(hipStream_t s, char transA, char transB, int m, int n, int k,
    double alpha, double const *A, int lda, double const *B, int ldb,
    double beta, double *C, int ldc)
{
    if (m <= 0 || n <= 0) return;
    if (k <= 0 || alpha == 0.0) { empty_product(s, m, n, beta, C, ldc); return; }
    bool ta = (transA == 'T' || transA == 't');
    bool tb = (transB == 'T' || transB == 't');
    if (ta && tb)        dispatch<true, true>(s, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
    else if (ta && !tb)  dispatch<true, false>(s, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
    else if (!ta && tb)  dispatch<false, true>(s, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
    else                 dispatch<false, false>(s, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
}

} // namespace sn
