// Host-array <-> HBM copies of the C-ABI shim (include/starneig/*.h take caller-owned, pageable,
// column-major arrays, like the reference: common/matrix.c:204-206 views the caller's memory).
// hipMemcpy2D on pageable memory moves ~25 GB/s on the MI355X box; here a few host threads each
// run a private two-slot pipeline (memcpy into a pinned slot | async DMA of the other slot), which
// keeps the PCIe Gen5 link busy (~50 GB/s).  starneig_node_enable_pinning() is accepted and changes
// nothing here: every copy goes through the pinned slots (registering a 3.2 GB array costs more than
// the extra host memcpy saves).  The lanes belong to the device that was current when they were
// created and are rebuilt when that changes; one API call at a time (the C interface is not
// re-entrant, SURVEY 8b "Threading").
#include "common.h"
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

namespace sn {

namespace {

constexpr size_t SLOT_BYTES = (size_t)16 << 20;
constexpr int MAX_THREADS = 8;

struct Lane {                       // one thread's pipeline
    double *slot[2] = {nullptr, nullptr};
    hipStream_t s = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
};
struct Stager {
    Lane lane[MAX_THREADS];
    int lanes = 0;
    int device = -1;                // the device the lanes' streams and events were created on
    void ensure(int n)
    {
        for (; lanes < n; lanes++) {
            Lane &l = lane[lanes];
            SN_HIP_CHECK(hipStreamCreateWithFlags(&l.s, hipStreamNonBlocking));
            for (int b = 0; b < 2; b++) {
                SN_HIP_CHECK(hipHostMalloc((void **)&l.slot[b], SLOT_BYTES, hipHostMallocDefault));
                SN_HIP_CHECK(hipEventCreateWithFlags(&l.ev[b], hipEventDisableTiming));
            }
        }
    }
    void release()
    {
        for (int k = 0; k < lanes; k++) {
            Lane &l = lane[k];
            for (int b = 0; b < 2; b++) { SN_HIP_CHECK(hipHostFree(l.slot[b])); SN_HIP_CHECK(hipEventDestroy(l.ev[b])); }
            SN_HIP_CHECK(hipStreamDestroy(l.s));
            l = Lane{};
        }
        lanes = 0;
    }
};
thread_local Stager g_stager;      // per host thread (one per device in the in-process multi-GPU path)

// columns [c0, c1) of a rows x cols matrix, chunk by chunk through the lane's two slots
void lane_copy(Lane &l, int device, bool to_device, double *dev, int ldd, double *host, int ldh,
    int rows, int c0, int c1)
{
    SN_HIP_CHECK(hipSetDevice(device));
    size_t const colbytes = (size_t)rows * 8;
    int const chunk = (int)std::max<size_t>(1, SLOT_BYTES / colbytes);
    int pending_c[2] = {0, 0}, pending_n[2] = {0, 0};
    int i = 0;
    auto drain = [&](int b) {       // device -> host: the slot's DMA is done, hand the columns to the caller
        if (pending_n[b] == 0) return;
        SN_HIP_CHECK(hipEventSynchronize(l.ev[b]));
        for (int c = 0; c < pending_n[b]; c++)
            std::memcpy(host + (size_t)(pending_c[b] + c) * ldh, l.slot[b] + (size_t)c * rows, colbytes);
        pending_n[b] = 0;
    };
    for (int c = c0; c < c1; c += chunk, i++) {
        int const b = i & 1, nc = std::min(chunk, c1 - c);
        if (to_device) {
            if (i >= 2) SN_HIP_CHECK(hipEventSynchronize(l.ev[b]));
            for (int k = 0; k < nc; k++)
                std::memcpy(l.slot[b] + (size_t)k * rows, host + (size_t)(c + k) * ldh, colbytes);
            SN_HIP_CHECK(hipMemcpy2DAsync(dev + (size_t)c * ldd, (size_t)ldd * 8, l.slot[b], colbytes, colbytes, nc,
                hipMemcpyHostToDevice, l.s));
            SN_HIP_CHECK(hipEventRecord(l.ev[b], l.s));
        } else {
            drain(b);
            SN_HIP_CHECK(hipMemcpy2DAsync(l.slot[b], colbytes, dev + (size_t)c * ldd, (size_t)ldd * 8, colbytes, nc,
                hipMemcpyDeviceToHost, l.s));
            SN_HIP_CHECK(hipEventRecord(l.ev[b], l.s));
            pending_c[b] = c; pending_n[b] = nc;
            drain(b ^ 1);
        }
    }
    if (!to_device) { drain(0); drain(1); }
    SN_HIP_CHECK(hipStreamSynchronize(l.s));
}

void staged_copy(bool to_device, double *dev, int ldd, double *host, int ldh, int rows, int cols, int threads)
{
    if (rows <= 0 || cols <= 0) return;
    size_t const colbytes = (size_t)rows * 8;
    if (colbytes > SLOT_BYTES) {    // (a column longer than a slot: n > 2 M rows -- not a size this path sees)
        SN_HIP_CHECK(hipMemcpy2D(to_device ? (void *)dev : (void *)host, (size_t)(to_device ? ldd : ldh) * 8,
            to_device ? (void *)host : (void *)dev, (size_t)(to_device ? ldh : ldd) * 8, colbytes, cols,
            to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
        return;
    }
    // small matrices: one lane; large ones: up to MAX_THREADS lanes over contiguous column ranges
    int const want = (int)std::min<size_t>(MAX_THREADS, std::max<size_t>(1, colbytes * cols / (4 * SLOT_BYTES)));
    int const T = std::max(1, std::min(threads, want));
    int device = 0;
    SN_HIP_CHECK(hipGetDevice(&device));
    if (g_stager.device != device) { g_stager.release(); g_stager.device = device; }
    g_stager.ensure(T);
    std::vector<std::thread> pool;
    int const per = divceil(cols, T);
    for (int t = 1; t < T; t++) {
        int const c0 = std::min(cols, t * per), c1 = std::min(cols, (t + 1) * per);
        if (c0 < c1) pool.emplace_back(lane_copy, std::ref(g_stager.lane[t]), device, to_device, dev, ldd, host, ldh, rows, c0, c1);
    }
    lane_copy(g_stager.lane[0], device, to_device, dev, ldd, host, ldh, rows, 0, std::min(cols, per));
    for (auto &th : pool) th.join();
}

} // namespace

// All copies are complete when the functions return (the shim's calls are blocking, like the reference's).
void upload_host_matrix(double *dev, int ldd, double const *host, int ldh, int rows, int cols, int threads)
{
    staged_copy(true, dev, ldd, const_cast<double *>(host), ldh, rows, cols, threads);
}
void download_host_matrix(double *host, int ldh, double const *dev, int ldd, int rows, int cols, int threads)
{
    staged_copy(false, const_cast<double *>(dev), ldd, host, ldh, rows, cols, threads);
}
void staging_release() { g_stager.release(); }

} // namespace sn
