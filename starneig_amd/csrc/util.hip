// Small device helpers: copies, fills, the reference test driver's LCG inputs,
// and the Frobenius-norm pieces of the acceptance checks.
#include "common.h"
#include "tuning.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace sn {

Tuning const &tuning()
{
    static Tuning const t = [] {
        Tuning t;
        char const *on = getenv("STARNEIG_AMD_TUNING");
        if (!on || atoi(on) == 0) return t;
        auto geti = [](char const *k, int d) { char const *v = getenv(k); return v ? atoi(v) : d; };
        auto getb = [](char const *k) { char const *v = getenv(k); return v != nullptr && strcmp(v, "0") != 0; };
        t.hess_wgs = geti("SN_HESS_WGS", t.hess_wgs);
        t.hess_max_split = geti("SN_HESS_MAXSPLIT", t.hess_max_split);
        t.hess_max_panels = geti("SN_HESS_MAX_PANELS", t.hess_max_panels);
        t.hess_cache_mb = geti("SN_HESS_CACHE_MB", (int)t.hess_cache_mb);
        t.hess_noside = getb("SN_HESS_NOSIDE");
        t.hess_fold = geti("SN_HESS_FOLD", t.hess_fold) == 2 ? 2 : 0;
        t.team_fail_rank = geti("SN_TEAM_FAIL_RANK", t.team_fail_rank);
        t.team_verify = getb("SN_TEAM_VERIFY");
        t.team_pooled_stream = getb("SN_TEAM_POOLED_STREAM");
        t.hess_side_cus = geti("SN_HESS_SIDE_CUS", 0);
        t.schur_nolazyrows = getb("SN_SCHUR_NOLAZYROWS");
        t.schur_lazy_batch = geti("SN_SCHUR_LAZY_BATCH", t.schur_lazy_batch);
        t.schur_helpers = geti("SN_SCHUR_HELPERS", t.schur_helpers);
        t.schur_reuse = std::max(0, std::min(8, geti("SN_SCHUR_REUSE", 0)));
        t.schur_nolookahead = getb("SN_SCHUR_NOLOOKAHEAD");
        t.schur_profile = getb("SN_SCHUR_PROFILE");
        t.aed_profile = getb("SN_AED_PROFILE");
        t.schur_cumask = geti("SN_SCHUR_CUMASK", t.schur_cumask);
        t.schur_hs_prio = geti("SN_SCHUR_HS_PRIO", 1) != 0;
        t.stream_mode = geti("SN_STREAM_MODE", t.stream_mode);
        t.stream_pad = geti("SN_STREAM_PAD", t.stream_pad);
        t.stream_space = getenv("SN_STREAM_SPACE");
        t.stream_pad_prio = geti("SN_STREAM_PAD_PRIO", t.stream_pad_prio);
        t.stream_lazy_free = geti("SN_STREAM_LAZY_FREE", t.stream_lazy_free);
        t.gemm_kchunk = geti("SN_GEMM_KCHUNK", t.gemm_kchunk);
        t.gemm_nosplit = getb("SN_GEMM_NOSPLIT");
        t.gemm_separate_sum = geti("SN_GEMM_SEPSUM", 1) != 0;
        t.ht_two_stage = geti("SN_HT_TWOSTAGE", t.ht_two_stage);
        t.ht2_min_n = geti("SN_HT2_MIN_N", t.ht2_min_n);
        t.gep_serial = getb("SN_GEP_SERIAL");
        t.gep_reuse = std::max(0, std::min(8, geti("SN_GEP_REUSE", 0)));
        t.gep_window = geti("SN_GEP_WINDOW", t.gep_window);
        return t;
    }();
    return t;
}

// A stream of the library.  critical: on the latency-bound chain of a reduction (high priority) -- else a
// bulk / lazy stream (priority `prio`).  The HIP runtime multiplexes the streams of one priority onto four
// hardware queues, whoever created them; a stream created with a CU mask gets a queue of its own.
void make_stream(hipStream_t *s, bool critical, int prio, int free_cus)
{
    static thread_local bool padded = false;
    if (!padded) {
        padded = true;
        int lo = 0, hi = 0;
        SN_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        for (int k = 0; k < tuning().stream_pad; k++) {
            hipStream_t d;
            int const level = tuning().stream_pad_prio;     // 0 high, 1 normal, 2 low
            SN_HIP_CHECK(hipStreamCreateWithPriority(&d, hipStreamNonBlocking, level == 0 ? hi : (level == 2 ? lo : (hi + lo) / 2)));
        }
    }
    // experiment: dummy queues of their own created before the k-th stream of this thread (SN_STREAM_SPACE=abcde...)
    static thread_local int created = 0;
    {
        char const *sp = tuning().stream_space;
        int const k = created++;
        if (sp && (int)strlen(sp) > k)
            for (int q = 0; q < sp[k] - '0'; q++) {
                hipDeviceProp_t prop; int dev = 0;
                SN_HIP_CHECK(hipGetDevice(&dev)); SN_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
                int const ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
                std::vector<uint32_t> mask(words, 0xffffffffu);
                hipStream_t d;
                SN_HIP_CHECK(hipExtStreamCreateWithCUMask(&d, words, mask.data()));
            }
    }
    int const mode = tuning().stream_mode;
    bool const dedicated = critical ? (mode & 1) : (mode & 2);
    if (dedicated) {
        hipDeviceProp_t prop; int dev = 0;
        SN_HIP_CHECK(hipGetDevice(&dev)); SN_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        int const ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
        std::vector<uint32_t> mask(words, 0u);
        for (int i = std::max(0, free_cus); i < ncu; i++) mask[i / 32] |= 1u << (i % 32);
        SN_HIP_CHECK(hipExtStreamCreateWithCUMask(s, words, mask.data()));
    } else
        SN_HIP_CHECK(hipStreamCreateWithPriority(s, hipStreamNonBlocking, prio));
}

__global__ void copy_matrix_kernel(int m, int n, double const *__restrict__ A, int lda,
    double *__restrict__ B, int ldb)
{
    int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    for (int c = blockIdx.y; c < n; c += gridDim.y)
        B[(size_t)c * ldb + r] = A[(size_t)c * lda + r];
}

void copy_matrix(hipStream_t s, int m, int n, double const *A, int lda, double *B, int ldb)
{
    if (m <= 0 || n <= 0) return;
    hipLaunchKernelGGL(copy_matrix_kernel, dim3(divceil(m, 256), std::min(n, 1024)), dim3(256),
        0, s, m, n, A, lda, B, ldb);
}

__global__ void set_matrix_kernel(int m, int n, double value, double diag,
    double *__restrict__ A, int lda)
{
    int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    for (int c = blockIdx.y; c < n; c += gridDim.y)
        A[(size_t)c * lda + r] = (r == c) ? diag : value;
}

void set_matrix(hipStream_t s, int m, int n, double value, double diag, double *A, int lda)
{
    if (m <= 0 || n <= 0) return;
    hipLaunchKernelGGL(set_matrix_kernel, dim3(divceil(m, 256), std::min(n, 1024)), dim3(256),
        0, s, m, n, value, diag, A, lda);
}

// x_{k} = a^k x_0 + c (a^k-1)/(a-1)  (mod 2^31): jump ahead by composing the
// affine map with itself (arithmetic mod 2^32, masked at the end of each step).
__device__ inline unsigned lcg_jump(unsigned x, unsigned long long k)
{
    unsigned a = 1103515245u, c = 12345u;     // test/common/common.c:58
    unsigned A = 1u, C = 0u;                  // identity map
    while (k) {
        if (k & 1ull) { C = a * C + c; A = a * A; }   // (A,C) <- (a,c) o (A,C)
        c = (a + 1u) * c;                             // (a,c) <- (a,c) o (a,c)
        a = a * a;
        k >>= 1;
    }
    return (A * x + C) & 0x7fffffffu;
}

constexpr int LCG_RUN = 64;   // consecutive elements per thread

__global__ void lcg_fill_kernel(int m, int n, unsigned seed, int mode,
    double *__restrict__ A, int lda)
{
    unsigned long long total = (unsigned long long)m * n;
    unsigned long long e0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * LCG_RUN;
    if (e0 >= total) return;
    unsigned x = lcg_jump(seed & 0x7fffffffu, e0);
    // note: the reference keeps the un-masked seed only for the very first step;
    // seeds used by the test driver are < 2^31, where both agree.
    int col = (int)(e0 / m), row = (int)(e0 % m);
    for (int t = 0; t < LCG_RUN && e0 + t < total; t++) {
        x = (x * 1103515245u + 12345u) & 0x7fffffffu;
        double v = (double)x / 2147483647.0;
        if (mode == 1) v = 2.0 * v - 1.0;
        A[(size_t)col * lda + row] = v;
        if (++row == m) { row = 0; col++; }
    }
}

void lcg_fill(hipStream_t s, int m, int n, unsigned seed, int mode, double *A, int lda)
{
    if (m <= 0 || n <= 0) return;
    unsigned long long total = (unsigned long long)m * n;
    unsigned long long threads = (total + LCG_RUN - 1) / LCG_RUN;
    unsigned blocks = (unsigned)((threads + 255) / 256);
    hipLaunchKernelGGL(lcg_fill_kernel, dim3(blocks), dim3(256), 0, s, m, n, seed, mode, A, lda);
}

// Column c of a structured LCG matrix holds cnt(c) = min(m, c + extra) draws (extra = 2:
// upper Hessenberg, 1: upper triangular; test/common/init.c:122-138,159-175), the rest of
// the column is zero.  One workgroup per column; `skip` draws precede the matrix.
__global__ void lcg_structured_kernel(int m, int n, unsigned seed, unsigned long long skip,
    int extra, double *__restrict__ A, int lda)
{
    int const c = blockIdx.x;
    // draws before column c: sum_{k<c} min(m, k + extra)
    long long const f = m - extra > 0 ? m - extra : 0;    // columns 0..f-1 hold fewer than m draws
    long long const cc = c < f ? c : f;
    unsigned long long const before =
        (unsigned long long)(cc * (cc - 1) / 2 + cc * extra + ((long long)c - cc) * m);
    int const cnt = min(m, c + extra);
    for (int r0 = threadIdx.x * LCG_RUN; r0 < m; r0 += blockDim.x * LCG_RUN) {
        int const rend = min(m, r0 + LCG_RUN);
        if (r0 < cnt) {
            unsigned x = lcg_jump(seed & 0x7fffffffu, skip + before + r0);
            for (int r = r0; r < rend; r++) {
                if (r < cnt) {
                    x = (x * 1103515245u + 12345u) & 0x7fffffffu;
                    A[(size_t)c * lda + r] = 2.0 * ((double)x / 2147483647.0) - 1.0;
                } else A[(size_t)c * lda + r] = 0.0;
            }
        } else
            for (int r = r0; r < rend; r++) A[(size_t)c * lda + r] = 0.0;
    }
}

static unsigned long long structured_draws(int m, int n, int extra)
{
    unsigned long long t = 0;
    for (int c = 0; c < n; c++) t += (unsigned long long)std::min(m, c + extra);
    return t;
}

void lcg_pencil(hipStream_t s, int n, unsigned seed, double *H, int ldh, double *R, int ldr)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(lcg_structured_kernel, dim3(n), dim3(256), 0, s, n, n, seed, 0ull, 2, H, ldh);
    hipLaunchKernelGGL(lcg_structured_kernel, dim3(n), dim3(256), 0, s, n, n, seed,
        structured_draws(n, n, 2), 1, R, ldr);
}

// acc[0] += sum (X - sub*Y)^2 (+ diag shift), acc[1] += count of nonzeros below sub-diagonal
__global__ void sumsq_diff_kernel(int m, int n, double const *__restrict__ X, int ldx,
    double const *__restrict__ Y, int ldy, double ident, double *acc)
{
    __shared__ double red[256];
    int r = blockIdx.x * 256 + threadIdx.x;
    double s = 0.0;
    if (r < m) {
        for (int c = blockIdx.y; c < n; c += gridDim.y) {
            double v = X[(size_t)c * ldx + r];
            if (Y) v -= Y[(size_t)c * ldy + r];
            if (r == c) v -= ident;
            s += v * v;
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(acc, red[0]);
}

__global__ void count_below_kernel(int n, double const *__restrict__ H, int ldh, double *acc)
{
    int r = blockIdx.x * 256 + threadIdx.x;
    double cnt = 0.0;
    if (r < n)
        for (int c = blockIdx.y; c < n && c + 2 <= r; c += gridDim.y)
            if (H[(size_t)c * ldh + r] != 0.0) cnt += 1.0;
    if (cnt != 0.0) atomicAdd(acc, cnt);
}

// Deterministic sum of squares (the deflation threshold u*||H||_F must be bit-identical on every
// GPU that reduces a replica of H, and from run to run): per-block partial sums into `part`
// (sumsq_ordered_parts(m, n) doubles), added up in a fixed order by one block.
__global__ void sumsq_parts_kernel(int m, int n, double const *__restrict__ X, int ldx,
    double *__restrict__ part)
{
    __shared__ double red[256];
    int r = blockIdx.x * 256 + threadIdx.x;
    double s = 0.0;
    if (r < m)
        for (int c = blockIdx.y; c < n; c += gridDim.y) {
            double v = X[(size_t)c * ldx + r];
            s += v * v;
        }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}
__global__ void sum_ordered_kernel(int count, double const *__restrict__ part, double *out)
{
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}
int sumsq_ordered_parts(int m, int n) { return divceil(m, 256) * std::min(n, 256); }
void sumsq_ordered(hipStream_t s, int m, int n, double const *X, int ldx, double *part, double *out)
{
    dim3 grid(divceil(m, 256), std::min(n, 256));
    hipLaunchKernelGGL(sumsq_parts_kernel, grid, dim3(256), 0, s, m, n, X, ldx, part);
    hipLaunchKernelGGL(sum_ordered_kernel, dim3(1), dim3(256), 0, s, (int)(grid.x * grid.y), part, out);
}

void sumsq_diff(hipStream_t s, int m, int n, double const *X, int ldx, double const *Y, int ldy,
    double ident, double *acc)
{
    hipLaunchKernelGGL(sumsq_diff_kernel, dim3(divceil(m, 256), std::min(n, 256)), dim3(256), 0, s,
        m, n, X, ldx, Y, ldy, ident, acc);
}
void count_below(hipStream_t s, int n, double const *H, int ldh, double *acc)
{
    hipLaunchKernelGGL(count_below_kernel, dim3(divceil(n, 256), std::min(n, 256)), dim3(256), 0, s,
        n, H, ldh, acc);
}

} // namespace sn
