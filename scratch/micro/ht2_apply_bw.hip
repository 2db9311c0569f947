// What bandwidth can the two HBM-bound applications of a stage-2 wavefront of the two-stage Hessenberg-triangular
// reduction reach?  (csrc/ht_twostage.hip: apply_left_chunk inside ht2_geng_left_kernel, ht2_apply_right_kernel;
// in the reduction at n = 12000 they move 3.7 and 2.6-3.2 TB/s of read + write, a float4 copy reaches 6.3.)
// Stand-alone: the access pattern of ONE wavefront -- steps k = 0 .. count-1, row / column blocks of 64 starting at
// p_k = 1 + 127 k (no alignment, as in the chase) -- with constant reflectors, in several mappings.
//   left : X(p:p+64, c:n) <- (I - tau v v^T) X      for two matrices
//   right: X(top:p+128, p:p+64) <- X (I - tau v v^T) for two matrices
//   hipcc --offload-arch=gfx950 -O3 -o ht2_apply_bw ht2_apply_bw.hip ; ./ht2_apply_bw [n=12000] [reps=20]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

template <int CTRL>
__device__ __forceinline__ double dpp(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double x)
{
    x += dpp<0x128>(x); x += dpp<0x124>(x); x += dpp<0x122>(x); x += dpp<0x121>(x);
    return x;
}
__device__ __forceinline__ double wave_sum(double x)
{
    x = row16_sum(x);
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
}

struct Job { int n, count; double *X[2]; int ld; double const *v; double tau; };
__host__ __device__ __forceinline__ int pos(int k) { return 1 + 127 * k; }

// ---- left, as in the library: 1024 threads, CHUNK columns a workgroup, 16 lanes x 4 CONSECUTIVE rows a column
template <int CHUNK, bool STRIDED>
__global__ __launch_bounds__(1024, 8) void left_group16(Job jb, int nchunk)
{
    int const idx = blockIdx.x, chunk = idx % nchunk, rest = idx / nchunk, k = rest % jb.count, z = rest / jb.count;
    int const p = pos(k), cb = z ? p : max(p - 63, 0);
    int const cbeg = cb + chunk * CHUNK;
    if (cbeg >= jb.n) return;
    double *X = jb.X[z];
    int const tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
    double v[4];
    #pragma unroll
    for (int q = 0; q < 4; q++) v[q] = jb.v[STRIDED ? l16 + 16 * q : 4 * l16 + q];
    constexpr int NC = CHUNK / 64;
    double y[NC][4], d[NC];
    #pragma unroll
    for (int u = 0; u < NC; u++) {
        int const c = cbeg + grp + 64 * u;
        double const *x = X + (size_t)c * jb.ld + p + (STRIDED ? l16 : 4 * l16);
        d[u] = 0.0;
        #pragma unroll
        for (int q = 0; q < 4; q++) { y[u][q] = (c < jb.n) ? x[STRIDED ? 16 * q : q] : 0.0; d[u] += v[q] * y[u][q]; }
    }
    #pragma unroll
    for (int u = 0; u < NC; u++) d[u] = row16_sum(d[u]) * jb.tau;
    #pragma unroll
    for (int u = 0; u < NC; u++) {
        int const c = cbeg + grp + 64 * u;
        double *x = X + (size_t)c * jb.ld + p + (STRIDED ? l16 : 4 * l16);
        #pragma unroll
        for (int q = 0; q < 4; q++) if (c < jb.n) x[STRIDED ? 16 * q : q] = y[u][q] - d[u] * v[q];
    }
}

// ---- left, lane = row: a wave takes 64 rows x CW columns (512 contiguous bytes an instruction), full-wave sums
template <int CW, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void left_lane_row(Job jb, int nchunk)
{
    int const idx = blockIdx.x, chunk = idx % nchunk, rest = idx / nchunk, k = rest % jb.count, z = rest / jb.count;
    int const p = pos(k), cb = z ? p : max(p - 63, 0);
    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int const cbeg = cb + (chunk * WAVES + wave) * CW;
    if (cbeg >= jb.n) return;
    double *X = jb.X[z] + (size_t)cbeg * jb.ld + p + lane;
    double const v = jb.v[lane];
    double y[CW], d[CW];
    #pragma unroll
    for (int u = 0; u < CW; u++) y[u] = (cbeg + u < jb.n) ? X[(size_t)u * jb.ld] : 0.0;
    #pragma unroll
    for (int u = 0; u < CW; u++) d[u] = wave_sum(v * y[u]) * jb.tau;
    #pragma unroll
    for (int u = 0; u < CW; u++) if (cbeg + u < jb.n) X[(size_t)u * jb.ld] = y[u] - d[u] * v;
}


// ---- left, column-major work order: a workgroup takes 16 columns x KB consecutive steps -- the steps of a wavefront
// are 127 rows apart, so along ONE column they are 512-byte pieces with 504-byte gaps: walked together they stay inside
// the same DRAM pages; the library's order (a workgroup = one step x 128 columns) touches 512 bytes a column and moves on
template <int KB>
__global__ __launch_bounds__(1024, 8) void left_colmajor(Job jb, int ncc, int nkb)
{
    int const idx = blockIdx.x, cc = idx % ncc, rest = idx / ncc, kb = rest % nkb, z = rest / nkb;
    int const tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
    int const c = cc * 16 + (grp & 15), slot = grp >> 4;
    double *X = jb.X[z];
    constexpr int NR = KB / 4;
    double y[NR][4], v[4], d[NR];
    #pragma unroll
    for (int q = 0; q < 4; q++) v[q] = jb.v[4 * l16 + q];
    bool live[NR];
    #pragma unroll
    for (int u = 0; u < NR; u++) {
        int const k = kb * KB + 4 * u + slot, p = pos(k), cb = z ? p : max(p - 63, 0);
        live[u] = k < jb.count && c >= cb && c < jb.n;
        double const *x = X + (size_t)min(c, jb.n - 1) * jb.ld + min(p, jb.n - 64) + 4 * l16;
        d[u] = 0.0;
        #pragma unroll
        for (int q = 0; q < 4; q++) { y[u][q] = live[u] ? x[q] : 0.0; d[u] += v[q] * y[u][q]; }
    }
    #pragma unroll
    for (int u = 0; u < NR; u++) d[u] = row16_sum(d[u]) * jb.tau;
    #pragma unroll
    for (int u = 0; u < NR; u++) {
        int const k = kb * KB + 4 * u + slot, p = pos(k);
        double *x = X + (size_t)min(c, jb.n - 1) * jb.ld + min(p, jb.n - 64) + 4 * l16;
        #pragma unroll
        for (int q = 0; q < 4; q++) if (live[u]) x[q] = y[u][q] - d[u] * v[q];
    }
}

// ---- right, as in the library: 256 threads, 64 rows x 64 columns, lane = row, wave = 16 columns, sums meet in LDS
__global__ __launch_bounds__(256) void right_lib(Job jb)
{
    __shared__ double s_d[4][64];
    int const k = blockIdx.y, z = blockIdx.z, p = pos(k);
    int const rows = min(p + 128, jb.n), top = 1, base = top & ~15;
    if (base + (int)blockIdx.x * 64 >= rows) return;
    int const lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    int const row = base + blockIdx.x * 64 + lane, q0 = 16 * part;
    bool const live = row < rows && row >= top;
    double *x = jb.X[z] + (size_t)(p + q0) * jb.ld + row;
    double y[16], v[16], d = 0.0;
    #pragma unroll
    for (int q = 0; q < 16; q++) { v[q] = jb.v[q0 + q]; y[q] = live ? x[(size_t)q * jb.ld] : 0.0; d += y[q] * v[q]; }
    s_d[part][lane] = d;
    __syncthreads();
    d = ((s_d[0][lane] + s_d[1][lane]) + (s_d[2][lane] + s_d[3][lane])) * jb.tau;
    #pragma unroll
    for (int q = 0; q < 16; q++) if (live) x[(size_t)q * jb.ld] = y[q] - d * v[q];
}

// ---- right: ROWS rows a workgroup in tiles of 64 (a wave = 16 columns of every tile: 16 * ROWS / 64 loads in flight)
template <int ROWS>
__global__ __launch_bounds__(256) void right_tall(Job jb)
{
    __shared__ double s_d[ROWS / 64][4][64];
    int const k = blockIdx.y, z = blockIdx.z, p = pos(k);
    int const rows = min(p + 128, jb.n), top = 1, base = top & ~15;
    if (base + (int)blockIdx.x * ROWS >= rows) return;
    int const lane = threadIdx.x & 63, part = threadIdx.x >> 6, q0 = 16 * part;
    constexpr int T = ROWS / 64;
    double y[T][16], v[16], d[T];
    #pragma unroll
    for (int q = 0; q < 16; q++) v[q] = jb.v[q0 + q];
    #pragma unroll
    for (int t = 0; t < T; t++) {
        int const row = base + blockIdx.x * ROWS + 64 * t + lane;
        bool const live = row < rows && row >= top;
        double const *x = jb.X[z] + (size_t)(p + q0) * jb.ld + row;
        d[t] = 0.0;
        #pragma unroll
        for (int q = 0; q < 16; q++) { y[t][q] = live ? x[(size_t)q * jb.ld] : 0.0; d[t] += y[t][q] * v[q]; }
        s_d[t][part][lane] = d[t];
    }
    __syncthreads();
    #pragma unroll
    for (int t = 0; t < T; t++) {
        int const row = base + blockIdx.x * ROWS + 64 * t + lane;
        bool const live = row < rows && row >= top;
        double *x = jb.X[z] + (size_t)(p + q0) * jb.ld + row;
        double const dd = ((s_d[t][0][lane] + s_d[t][1][lane]) + (s_d[t][2][lane] + s_d[t][3][lane])) * jb.tau;
        #pragma unroll
        for (int q = 0; q < 16; q++) if (live) x[(size_t)q * jb.ld] = y[t][q] - dd * v[q];
    }
}

// ---- right: one wave = 64 rows x all 64 columns in four batches of 16, no LDS, no barrier; WAVES waves a workgroup
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void right_wave(Job jb)
{
    int const k = blockIdx.y, z = blockIdx.z, p = pos(k);
    int const rows = min(p + 128, jb.n), top = 1, base = top & ~15;
    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int const row = base + (blockIdx.x * WAVES + wave) * 64 + lane;
    if (row - lane >= rows) return;
    bool const live = row < rows && row >= top;
    double *x = jb.X[z] + (size_t)p * jb.ld + row;
    double y[64], d = 0.0;
    #pragma unroll
    for (int q = 0; q < 64; q++) { y[q] = live ? x[(size_t)q * jb.ld] : 0.0; }
    #pragma unroll
    for (int q = 0; q < 64; q++) d += y[q] * jb.v[q];
    d *= jb.tau;
    #pragma unroll
    for (int q = 0; q < 64; q++) if (live) x[(size_t)q * jb.ld] = y[q] - d * jb.v[q];
}

// ---- reference: in-place scale of the same bytes as ONE contiguous sweep (what a plain streaming kernel gets)
__global__ __launch_bounds__(256) void stream_inplace(double *x, size_t count)
{
    size_t const stride = (size_t)gridDim.x * 256 * 2;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2; i + 1 < count; i += stride) {
        double2 v = *reinterpret_cast<double2 *>(x + i);
        v.x *= 1.0000001; v.y *= 1.0000001;
        *reinterpret_cast<double2 *>(x + i) = v;
    }
}

int main(int argc, char **argv)
{
    int const n = argc > 1 ? atoi(argv[1]) : 12000, reps = argc > 2 ? atoi(argv[2]) : 20;
    int const ld = (n + 15) / 16 * 16;
    double *A, *B, *v;
    CK(hipMalloc(&A, (size_t)ld * n * 8)); CK(hipMalloc(&B, (size_t)ld * n * 8)); CK(hipMalloc(&v, 64 * 8));
    std::vector<double> hv(64, 0.05), init((size_t)ld * n, 1.0);
    CK(hipMemcpy(v, hv.data(), 64 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(A, init.data(), init.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(B, init.data(), init.size() * 8, hipMemcpyHostToDevice));
    int count = 0;
    while (pos(count) + 64 <= n - 1) count++;
    Job jb{n, count, {A, B}, ld, v, 0.1};
    double left_bytes = 0.0, right_bytes = 0.0;
    for (int k = 0; k < count; k++) {
        int const p = pos(k);
        left_bytes += 2.0 * 8.0 * 64.0 * ((n - p) + (n - (p > 63 ? p - 63 : 0)));
        right_bytes += 2.0 * 2.0 * 8.0 * 64.0 * ((p + 128 < n ? p + 128 : n) - 1);
    }
    printf("n = %d, %d steps a wavefront: left %.0f MB, right %.0f MB of read + write\n", n, count, left_bytes / 1e6, right_bytes / 1e6);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](char const *name, double bytes, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %-58s %8.1f us  %6.2f TB/s\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
    };
    timed("in-place stream over as many bytes as the left pass", left_bytes, [&] {
        hipLaunchKernelGGL(stream_inplace, dim3(4096), dim3(256), 0, 0, A, (size_t)(left_bytes / 16)); });
    {
        int nchunk = (n + 127) / 128;
        timed("left: library (1024 thr, 128 cols, 16 lanes x 4 consecutive)", left_bytes, [&] {
            hipLaunchKernelGGL((left_group16<128, false>), dim3(count * 2 * nchunk), dim3(1024), 0, 0, jb, nchunk); });
        timed("left: 16 lanes x 4 STRIDED rows (128 B an instruction/col)", left_bytes, [&] {
            hipLaunchKernelGGL((left_group16<128, true>), dim3(count * 2 * nchunk), dim3(1024), 0, 0, jb, nchunk); });
        nchunk = (n + 255) / 256;
        timed("left: 256 cols a workgroup, consecutive", left_bytes, [&] {
            hipLaunchKernelGGL((left_group16<256, false>), dim3(count * 2 * nchunk), dim3(1024), 0, 0, jb, nchunk); });
        timed("left: 256 cols a workgroup, strided", left_bytes, [&] {
            hipLaunchKernelGGL((left_group16<256, true>), dim3(count * 2 * nchunk), dim3(1024), 0, 0, jb, nchunk); });
        nchunk = (n + 4 * 8 - 1) / (4 * 8);
        timed("left: lane = row, wave = 8 cols, 4 waves", left_bytes, [&] {
            hipLaunchKernelGGL((left_lane_row<8, 4>), dim3(count * 2 * nchunk), dim3(256), 0, 0, jb, nchunk); });
        nchunk = (n + 4 * 16 - 1) / (4 * 16);
        timed("left: lane = row, wave = 16 cols, 4 waves", left_bytes, [&] {
            hipLaunchKernelGGL((left_lane_row<16, 4>), dim3(count * 2 * nchunk), dim3(256), 0, 0, jb, nchunk); });
        nchunk = (n + 16 * 16 - 1) / (16 * 16);
        timed("left: lane = row, wave = 16 cols, 16 waves", left_bytes, [&] {
            hipLaunchKernelGGL((left_lane_row<16, 16>), dim3(count * 2 * nchunk), dim3(1024), 0, 0, jb, nchunk); });
        {
            int const ncc = (n + 15) / 16;
            int nkb = (count + 7) / 8;
            timed("left: column-major, 16 cols x 8 steps a workgroup", left_bytes, [&] {
                hipLaunchKernelGGL(left_colmajor<8>, dim3(ncc * nkb * 2), dim3(1024), 0, 0, jb, ncc, nkb); });
            nkb = (count + 15) / 16;
            timed("left: column-major, 16 cols x 16 steps a workgroup", left_bytes, [&] {
                hipLaunchKernelGGL(left_colmajor<16>, dim3(ncc * nkb * 2), dim3(1024), 0, 0, jb, ncc, nkb); });
            nkb = (count + 3) / 4;
            timed("left: column-major, 16 cols x 4 steps a workgroup", left_bytes, [&] {
                hipLaunchKernelGGL(left_colmajor<4>, dim3(ncc * nkb * 2), dim3(1024), 0, 0, jb, ncc, nkb); });
        }
    }
    {
        timed("right: library (256 thr, 64 x 64, wave = 16 cols, LDS)", right_bytes, [&] {
            hipLaunchKernelGGL(right_lib, dim3((n + 63) / 64, count, 2), dim3(256), 0, 0, jb); });
        timed("right: 128 rows a workgroup", right_bytes, [&] {
            hipLaunchKernelGGL(right_tall<128>, dim3((n + 127) / 128, count, 2), dim3(256), 0, 0, jb); });
        timed("right: 256 rows a workgroup", right_bytes, [&] {
            hipLaunchKernelGGL(right_tall<256>, dim3((n + 255) / 256, count, 2), dim3(256), 0, 0, jb); });
        timed("right: wave = 64 rows x 64 cols, 1 wave a workgroup", right_bytes, [&] {
            hipLaunchKernelGGL(right_wave<1>, dim3((n + 63) / 64, count, 2), dim3(64), 0, 0, jb); });
        timed("right: wave = 64 rows x 64 cols, 4 waves a workgroup", right_bytes, [&] {
            hipLaunchKernelGGL(right_wave<4>, dim3((n + 255) / 256, count, 2), dim3(256), 0, 0, jb); });
    }
    return 0;
}
