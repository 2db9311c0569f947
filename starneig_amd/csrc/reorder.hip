// Eigenvalue reordering of a real Schur form on one MI355X (row f3 of SURVEY 8f).
//
// Reference: reorder/interface.c:210-263 (starneig_SEP_SM_ReorderSchur), reorder/core.c (chains
// of overlapping diagonal windows), reorder/cpu.c + cuda.cu:126-761 (window kernels: sequences of
// dtrexc-style swaps), common/cpu.c:54-162 (off-diagonal GEMM updates).  Kept: the algorithm --
// selected diagonal blocks travel to the top-left corner in groups, window by window; inside a
// window the selected blocks are swapped to its top by exact-arithmetic-equivalent orthogonal
// exchanges (Sylvester equation + QR, schur_host.hip swap_blocks, LAPACK dlaexc) accumulated
// into a small factor Z; everything outside the window sees Z through one GEMM per side.
// Re-designed for the GPU: S and Q stay in HBM; windows (<= 128 rows) are copied to pinned host
// memory, reordered there (the swaps are a chain of dependent 2x2..4x4 problems -- one CPU core
// beats one workgroup at that; several windows of a round go to several cores), and the three
// off-diagonal updates
//     S(0:wb, window) <- . Z,   S(window, we:n) <- Z^T .,   Q(:, window) <- . Z
// run as batched in-place fp64-MFMA tiles (dgemm_tile.h; no scratch copy, unlike common/tasks.c:459-462).
#include "common.h"
#include "schur_host.h"
#include <vector>
#include <algorithm>
#include <cmath>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <atomic>
#include <starneig/error.h>

namespace sn {

namespace {

constexpr int RW_MAX = 128;         // window rows: the in-place update tiles own a whole window
constexpr int RW_LD = RW_MAX + 8;   // leading dimension of a window's T / Z block in the staging buffers: an odd number of
                                    // cache lines (128 doubles are 16 lines: the row walks of the host swaps would live in 4 L1 sets)

struct WinMeta { int wb, w; };

// S(wb:wb+w, wb:wb+w) -> T[k] (dir = 0) or back (dir = 1), one workgroup column per window column
__global__ __launch_bounds__(128) void reorder_copy_windows_kernel(WinMeta const *__restrict__ meta,
    double *__restrict__ S, int lds, double *__restrict__ T, int dir)
{
    WinMeta const m = meta[blockIdx.y];
    int const j = blockIdx.x, i = threadIdx.x;
    if (j >= m.w || i >= m.w) return;
    double *g = S + (size_t)(m.wb + j) * lds + m.wb + i;
    double *t = T + ((size_t)blockIdx.y * RW_LD + j) * RW_LD + i;
    if (dir == 0) *t = *g; else *g = *t;
}

// a few host threads that reorder the windows of one round side by side.  A round's state (job,
// total, next) is only ever written while no worker is inside work(): run() returns when every
// index is done AND every worker that joined the round has left it (`active`), so a worker that
// fetched a surplus index late cannot meet the next round's counters.
class WindowPool {
public:
    explicit WindowPool(int threads)
    {
        for (int t = 0; t < threads; t++) workers.emplace_back([this] { loop(); });
    }
    ~WindowPool()
    {
        { std::lock_guard<std::mutex> l(m); quit = true; round++; }
        cv.notify_all();
        for (auto &w : workers) w.join();
    }
    template <typename F> void run(int count, F const &f)
    {
        if (workers.empty() || count == 1) { for (int k = 0; k < count; k++) f(k); return; }
        {
            std::lock_guard<std::mutex> l(m);
            job = [&f](int k) { f(k); };
            total = count; next.store(0); done = 0; round++;
        }
        cv.notify_all();
        work(count);                                            // the caller takes its share
        std::unique_lock<std::mutex> l(m);
        cv_done.wait(l, [&] { return done >= total && active == 0; });
        total = 0;          // a worker that wakes up from here on has nothing to join
    }
private:
    void work(int total_)
    {
        int mine = 0;
        for (;;) {
            int const k = next.fetch_add(1);
            if (k >= total_) break;
            job(k); mine++;
        }
        std::lock_guard<std::mutex> l(m);
        done += mine;
    }
    void loop()
    {
        long seen = 0;
        for (;;) {
            int total_;
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [&] { return round != seen; });
                seen = round;
                if (quit) return;
                if (total == 0) continue;                       // the round ended before this thread woke up
                total_ = total; active++;
            }
            work(total_);
            { std::lock_guard<std::mutex> l(m); active--; }
            cv_done.notify_all();
        }
    }
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv, cv_done;
    std::function<void(int)> job;
    std::atomic<int> next{0};
    int total = 0, done = 0, active = 0;
    long round = 0;
    bool quit = false;
};

struct ReorderWorkspace {
    double *hT = nullptr, *hZ = nullptr, *hDiag = nullptr;      // pinned
    double *dT = nullptr, *dZ = nullptr;
    WinMeta *hMeta = nullptr, *dMeta = nullptr;
    GemmDesc *hDesc = nullptr, *dDesc = nullptr;
    int n = 0, cap = 0;                                         // cap: windows of one round
    hipStream_t s = nullptr;
    hipEvent_t fence = nullptr, done = nullptr;
    void ensure(int n_)
    {
        if (!s) {
            SN_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            SN_HIP_CHECK(hipEventCreateWithFlags(&fence, hipEventDisableTiming));
            SN_HIP_CHECK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        }
        if (n_ > n) {
            release();
            n = n_;
            cap = std::max(4, divceil(n, RW_MAX / 2) + 2);
            size_t const blk = (size_t)RW_LD * RW_LD * sizeof(double);
            SN_HIP_CHECK(hipHostMalloc((void **)&hDiag, (size_t)3 * n * 8, hipHostMallocDefault));
            SN_HIP_CHECK(hipHostMalloc((void **)&hT, cap * blk, hipHostMallocDefault));
            SN_HIP_CHECK(hipHostMalloc((void **)&hZ, cap * blk, hipHostMallocDefault));
            SN_HIP_CHECK(hipHostMalloc((void **)&hMeta, cap * sizeof(WinMeta), hipHostMallocDefault));
            SN_HIP_CHECK(hipHostMalloc((void **)&hDesc, (size_t)3 * cap * sizeof(GemmDesc), hipHostMallocDefault));
            SN_HIP_CHECK(hipMalloc((void **)&dT, cap * blk));
            SN_HIP_CHECK(hipMalloc((void **)&dZ, cap * blk));
            SN_HIP_CHECK(hipMalloc((void **)&dMeta, cap * sizeof(WinMeta)));
            SN_HIP_CHECK(hipMalloc((void **)&dDesc, (size_t)3 * cap * sizeof(GemmDesc)));
        }
    }
    void release()
    {
        void *host[] = {hT, hZ, hDiag, hMeta, hDesc}; void *dev[] = {dT, dZ, dMeta, dDesc};
        for (void *p : host) if (p) SN_HIP_CHECK(hipHostFree(p));
        for (void *p : dev) if (p) SN_HIP_CHECK(hipFree(p));
        hT = hZ = hDiag = dT = dZ = nullptr; hMeta = dMeta = nullptr; hDesc = dDesc = nullptr;
        n = 0;
    }
};
ReorderWorkspace g_rws;

} // namespace

void reorder_release_workspace() { g_rws.release(); }

// selected: HOST array of n marks (in: selected eigenvalues; out: final positions of the correctly
// placed ones).  real/imag: HOST arrays (may be NULL).  window_size / values_per_chain <= 0 select
// the defaults (128 rows, half a window).  stats (may be NULL): [0] windows, [1] executed GEMM flops,
// [2] rounds.
//
// Rounds of disjoint windows (the reference's chains of windows, reorder/core.c, as a wavefront):
// every round the diagonal is cut, from the top, into windows that each hold at most
// values_per_chain rows of selected blocks and the unselected rows above them; all windows of a
// round are downloaded in one copy, reordered side by side by `host_threads` threads, uploaded
// in one copy, and their three off-diagonal updates run as three batched launches (right
// updates of S, left updates of S, right updates of Q -- left and right factors of different
// windows act on disjoint rows / columns and commute).  A group of selected blocks climbs one
// window per round; the groups below follow in the rows it has left.
int reorder_schur_device(hipStream_t caller, int n, int *selected, double *dS, int ldS,
    double *dQ, int ldQ, double *real, double *imag, int window_size, int values_per_chain,
    double *stats, int host_threads)
{
    ReorderWorkspace &ws = g_rws;
    ws.ensure(n);
    hipStream_t s = ws.s;
    SN_HIP_CHECK(hipEventRecord(ws.fence, caller));
    SN_HIP_CHECK(hipStreamWaitEvent(s, ws.fence, 0));

    int const W = std::max(8, std::min(RW_MAX, window_size > 0 ? window_size : RW_MAX));
    int const kmax = std::max(1, std::min(W - 2, values_per_chain > 0 ? values_per_chain : W / 2));
    double windows = 0.0, flops = 0.0, rounds = 0.0;

    // block structure: the sub-diagonal of S
    double *sub = ws.hDiag;                                 // sub[i] = S(i+1, i)
    if (n > 1) {
        SN_HIP_CHECK(hipMemcpy2DAsync(sub, 8, dS + 1, (size_t)(ldS + 1) * 8, 8, n - 1, hipMemcpyDeviceToHost, s));
        SN_HIP_CHECK(hipStreamSynchronize(s));
    }
    std::vector<int> sel(selected, selected + n);
    auto is_pair = [&](int i) { return i + 1 < n && sub[i] != 0.0; };
    for (int i = 0; i < n; i++) {
        sel[i] = sel[i] != 0;
        if (is_pair(i)) { sel[i] = sel[i + 1] = (sel[i] || selected[i + 1] != 0); i++; }
    }

    WindowPool pool(std::max(0, std::min(host_threads, 32) - 1));
    std::vector<int> placed_of(ws.cap), failed_of(ws.cap);
    size_t const blk = (size_t)RW_LD * RW_LD;
    int rc = STARNEIG_SUCCESS;
    while (rc == STARNEIG_SUCCESS) {
        // ---- the windows of this round
        int dest = 0;
        while (dest < n && sel[dest]) dest++;               // rows [0, dest) hold placed selected blocks
        int count = 0, p = dest;
        while (count < ws.cap) {
            int f = p;
            while (f < n && !sel[f]) f++;
            if (f >= n) break;
            // selected blocks from f downwards: at most kmax rows of them, inside one window
            int we = f, cnt = 0;
            while (we < n && we - f < W) {
                int const bs = is_pair(we) ? 2 : 1;
                if (we + bs - f > W) break;
                // (the first selected block of a window is always admitted: with values_per_chain = 1 a
                // selected 2x2 block would otherwise never move and the round loop never end)
                if (sel[we]) { if (cnt > 0 && cnt + bs > kmax) break; cnt += bs; }
                we += bs;
            }
            while (we > f && !sel[we - 1]) we -= (we - 2 >= f && is_pair(we - 2)) ? 2 : 1;   // unselected rows at the bottom
            int wb = std::max(p, we - W);
            if (wb > 0 && sub[wb - 1] != 0.0) wb++;         // do not cut a 2x2 block
            // nothing to do when the window starts with its selected rows and holds nothing else
            bool work = false;
            for (int i = wb, seen_unsel = 0; i < we && !work; i++) { if (!sel[i]) seen_unsel = 1; else if (seen_unsel) work = true; }
            if (work && we - wb >= 2) { ws.hMeta[count] = WinMeta{wb, we - wb}; count++; }
            p = we;
        }
        if (count == 0) break;
        rounds += 1.0; windows += count;
        // ---- windows -> host
        SN_HIP_CHECK(hipMemcpyAsync(ws.dMeta, ws.hMeta, count * sizeof(WinMeta), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(reorder_copy_windows_kernel, dim3(W, count), dim3(128), 0, s, ws.dMeta, dS, ldS, ws.dT, 0);
        SN_HIP_CHECK(hipMemcpyAsync(ws.hT, ws.dT, count * blk * 8, hipMemcpyDeviceToHost, s));
        SN_HIP_CHECK(hipStreamSynchronize(s));
        // ---- reorder them
        pool.run(count, [&](int k) {
            int const wb = ws.hMeta[k].wb, w = ws.hMeta[k].w;
            double *T = ws.hT + k * blk, *Z = ws.hZ + k * blk;
            for (int j = 0; j < w; j++)
                for (int i = 0; i < w; i++) Z[(size_t)j * RW_LD + i] = (i == j) ? 1.0 : 0.0;
            int failed = 0;
            placed_of[k] = host::reorder_window(w, T, RW_LD, Z, RW_LD, sel.data() + wb, &failed);
            failed_of[k] = failed;
            for (int i = 0; i + 1 < w; i++) sub[wb + i] = T[(size_t)i * RW_LD + i + 1];
        });
        // ---- back to the device; off-diagonal updates
        SN_HIP_CHECK(hipMemcpyAsync(ws.dT, ws.hT, count * blk * 8, hipMemcpyHostToDevice, s));
        SN_HIP_CHECK(hipMemcpyAsync(ws.dZ, ws.hZ, count * blk * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(reorder_copy_windows_kernel, dim3(W, count), dim3(128), 0, s, ws.dMeta, dS, ldS, ws.dT, 1);
        int nr = 0, nl = 0, nq = 0, max_r = 0, max_l = 0;
        GemmDesc *dr = ws.hDesc, *dl = ws.hDesc + ws.cap, *dq = ws.hDesc + 2 * ws.cap;
        for (int k = 0; k < count; k++) {
            int const wb = ws.hMeta[k].wb, w = ws.hMeta[k].w, we = wb + w;
            double const *Z = ws.dZ + k * blk;
            if (wb > 0) { double *X = dS + (size_t)wb * ldS; dr[nr++] = GemmDesc{X, Z, X, wb, w, w, ldS, RW_LD, ldS}; max_r = std::max(max_r, wb); }
            if (n - we > 0) { double *X = dS + (size_t)we * ldS + wb; dl[nl++] = GemmDesc{Z, X, X, w, n - we, w, RW_LD, ldS, ldS}; max_l = std::max(max_l, n - we); }
            if (dQ) { double *X = dQ + (size_t)wb * ldQ; dq[nq++] = GemmDesc{X, Z, X, n, w, w, ldQ, RW_LD, ldQ}; }
            flops += 2.0 * w * w * ((double)wb + (n - we) + (dQ ? n : 0));
            if (failed_of[k]) rc = STARNEIG_PARTIAL_REORDERING;
        }
        SN_HIP_CHECK(hipMemcpyAsync(ws.dDesc, ws.hDesc, (size_t)3 * ws.cap * sizeof(GemmDesc), hipMemcpyHostToDevice, s));
        dgemm_batched_right_inplace(s, ws.dDesc, nr, max_r);
        dgemm_batched_left_inplace(s, ws.dDesc + ws.cap, nl, max_l);
        dgemm_batched_right_inplace(s, ws.dDesc + 2 * ws.cap, nq, n);
        // the staging buffers are reused by the next round
        SN_HIP_CHECK(hipStreamSynchronize(s));
    }
    // marks of the correctly placed eigenvalues (reorder/interface.c:166-187)
    {
        int placed = 0;
        while (placed < n && sel[placed]) placed++;
        for (int i = 0; i < n; i++) selected[i] = i < placed ? 1 : 0;
    }

    // eigenvalues from the diagonal blocks (common/tasks.c:1113-1166)
    if (real && imag) {
        double *dg = ws.hDiag + n, *sup = ws.hDiag + 2 * n;
        SN_HIP_CHECK(hipMemcpy2DAsync(dg, 8, dS, (size_t)(ldS + 1) * 8, 8, n, hipMemcpyDeviceToHost, s));
        if (n > 1) {
            SN_HIP_CHECK(hipMemcpy2DAsync(sub, 8, dS + 1, (size_t)(ldS + 1) * 8, 8, n - 1, hipMemcpyDeviceToHost, s));
            SN_HIP_CHECK(hipMemcpy2DAsync(sup, 8, dS + ldS, (size_t)(ldS + 1) * 8, 8, n - 1, hipMemcpyDeviceToHost, s));
        }
        SN_HIP_CHECK(hipStreamSynchronize(s));
        for (int i = 0; i < n; i++) {
            if (i + 1 < n && sub[i] != 0.0) {
                double a = dg[i], b = sup[i], c = sub[i], d = dg[i + 1], cs, sn_;
                host::lanv2(a, b, c, d, real[i], imag[i], real[i + 1], imag[i + 1], cs, sn_);
                i++;
            } else { real[i] = dg[i]; imag[i] = 0.0; }
        }
    }
    SN_HIP_CHECK(hipEventRecord(ws.done, s));
    SN_HIP_CHECK(hipStreamWaitEvent(caller, ws.done, 0));
    SN_HIP_CHECK(hipStreamSynchronize(s));
    if (stats) { stats[0] = windows; stats[1] = flops; stats[2] = rounds; }
    return rc;
}

} // namespace sn
