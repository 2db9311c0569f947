// Small device helpers: copies, fills, the reference test driver's LCG inputs,
// and the Frobenius-norm pieces of the acceptance checks.
#include "common.h"

namespace sn {

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
