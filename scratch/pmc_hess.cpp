// One Hessenberg reduction through the device C-ABI from a single-threaded C++ program, no Python / torch in the
// process: rocprofv3 --pmc with SQ counters segfaults in the profiler's dispatch hook over the Python drivers at
// n = 20000; usage: pmc_hess [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <algorithm>
#include "starneig/node.h"
#include "starneig_amd.h"
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char **argv)
{
    int const n = argc > 1 ? atoi(argv[1]) : 20000;
    int const ld = (n + 15) / 16 * 16;
    starneig_node_init(1, 1, STARNEIG_NO_MESSAGES);
    double *dA, *dQ;
    CHECK(hipMalloc((void **)&dA, (size_t)ld * n * 8)); CHECK(hipMalloc((void **)&dQ, (size_t)ld * n * 8));
    if (!getenv("PMC_NO_MEMSET")) { CHECK(hipMemset(dA, 0, (size_t)ld * n * 8)); CHECK(hipMemset(dQ, 0, (size_t)ld * n * 8)); }
    // filled in column blocks: the profiler's counter mode dies on launches of 2^23 work-items or more
    // (n = 8000: the 32 x 1024 x 256 fill launch), so no launch of the set-up may be that large
    int const cb = std::max(64, (1 << 21) / ((n + 255) / 256 * 256) * 1);     // columns per block: <= 2^21 entries... x 1 thread each
    for (int c0 = 0; c0 < n; c0 += cb) {
        int const w = std::min(cb, n - c0);
        starneig_amd_lcg_fill_device(n, w, 2019u + (unsigned)c0, 0, dA + (size_t)c0 * ld, ld, nullptr);
        starneig_amd_set_matrix_device(n, w, 0.0, 0.0, dQ + (size_t)c0 * ld, ld, nullptr);
        starneig_amd_set_matrix_device(w, w, 0.0, 1.0, dQ + (size_t)c0 * ld + c0, ld, nullptr);
    }
    CHECK(hipDeviceSynchronize());
    double stats[16] = {0};
    auto t0 = std::chrono::steady_clock::now();
    int rc = starneig_amd_hessenberg_device(n, 0, n, -1, dA, ld, dQ, ld, nullptr, stats);
    CHECK(hipDeviceSynchronize());
    double const dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("pmc_hess n=%d rc=%d %.3f s\n", n, rc, dt);
    starneig_node_finalize();
    return rc;
}
