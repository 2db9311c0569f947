// Eigenvalue reordering of a real Schur form on one MI355X (row f3 of SURVEY 8f).
//
// Reference: reorder/interface.c:210-263 (starneig_SEP_SM_ReorderSchur), reorder/core.c (chains
// of overlapping diagonal windows), reorder/cpu.c + cuda.cu:126-761 (window kernels: sequences of
// dtrexc-style swaps), common/cpu.c:54-162 (off-diagonal GEMM updates).  Kept: the algorithm --
// selected diagonal blocks travel to the top-left corner in groups, window by window; inside a
// window the selected blocks are swapped to its top by exact-arithmetic-equivalent orthogonal
// exchanges (Sylvester equation + QR, schur_host.hip swap_blocks, LAPACK dlaexc) accumulated
// into a small factor Z; everything outside the window sees Z through one GEMM per side.
// Re-designed for the GPU: S and Q stay in HBM; a window (<= 128 rows) is copied to pinned host
// memory, reordered there (the swaps are a chain of dependent 2x2..4x4 problems -- one CPU core
// beats one workgroup at that), and the three off-diagonal updates
//     S(0:wb, window) <- . Z,   S(window, we:n) <- Z^T .,   Q(:, window) <- . Z
// run as in-place fp64-MFMA tiles (dgemm_tile.h; no scratch copy, unlike common/tasks.c:459-462).
#include "common.h"
#include "schur_host.h"
#include <vector>
#include <algorithm>
#include <cmath>
#include <starneig/error.h>

namespace sn {

namespace {

constexpr int RW_MAX = 128;         // window rows: the in-place update tiles own a whole window

struct ReorderWorkspace {
    double *hT = nullptr, *hZ = nullptr, *hDiag = nullptr;      // pinned
    double *dZ = nullptr;
    int n = 0;
    hipStream_t s = nullptr;
    hipEvent_t fence = nullptr, done = nullptr;
    void ensure(int n_) {
        if (!s) {
            SN_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            SN_HIP_CHECK(hipEventCreateWithFlags(&fence, hipEventDisableTiming));
            SN_HIP_CHECK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
            SN_HIP_CHECK(hipHostMalloc((void **)&hT, (size_t)(RW_MAX + 8) * RW_MAX * 8, hipHostMallocDefault));
            SN_HIP_CHECK(hipHostMalloc((void **)&hZ, (size_t)(RW_MAX + 8) * RW_MAX * 8, hipHostMallocDefault));
            SN_HIP_CHECK(hipMalloc((void **)&dZ, (size_t)RW_MAX * RW_MAX * 8));
        }
        if (n_ > n) {
            if (hDiag) SN_HIP_CHECK(hipHostFree(hDiag));
            SN_HIP_CHECK(hipHostMalloc((void **)&hDiag, (size_t)3 * n_ * 8, hipHostMallocDefault));
            n = n_;
        }
    }
    void release() {
        if (hDiag) { SN_HIP_CHECK(hipHostFree(hDiag)); hDiag = nullptr; }
        n = 0;
    }
};
ReorderWorkspace g_rws;

} // namespace

void reorder_release_workspace() { g_rws.release(); }

// selected: HOST array of n marks (in: selected eigenvalues; out: final positions of the correctly
// placed ones).  real/imag: HOST arrays (may be NULL).  window_size / values_per_chain <= 0 select
// the defaults (128 rows, half a window).  stats (may be NULL): [0] windows, [1] executed GEMM flops.
int reorder_schur_device(hipStream_t caller, int n, int *selected, double *dS, int ldS,
    double *dQ, int ldQ, double *real, double *imag, int window_size, int values_per_chain,
    double *stats)
{
    ReorderWorkspace &ws = g_rws;
    ws.ensure(n);
    hipStream_t s = ws.s;
    SN_HIP_CHECK(hipEventRecord(ws.fence, caller));
    SN_HIP_CHECK(hipStreamWaitEvent(s, ws.fence, 0));

    int const W = std::max(8, std::min(RW_MAX, window_size > 0 ? window_size : RW_MAX));
    int const kmax = std::max(1, std::min(W - 2, values_per_chain > 0 ? values_per_chain : W / 2));
    int const ldh = W + 8;
    double windows = 0.0, flops = 0.0;

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

    int rc = STARNEIG_SUCCESS;
    int dest = 0;                                           // rows [0, dest) hold placed selected blocks
    while (rc == STARNEIG_SUCCESS) {
        while (dest < n && sel[dest]) dest++;
        int f = dest;
        while (f < n && !sel[f]) f++;
        if (f >= n) break;                                  // nothing selected below dest
        // the group: selected blocks from f downwards, at most kmax rows of them, inside one window
        int we = f, cnt = 0;
        while (we < n && we - f < W) {
            int const bs = is_pair(we) ? 2 : 1;
            if (we + bs - f > W) break;
            if (sel[we]) { if (cnt + bs > kmax) break; cnt += bs; }
            we += bs;
        }
        // drop unselected blocks at the bottom of the group region
        while (we > f && !sel[we - 1]) we -= (we - 2 >= f && is_pair(we - 2)) ? 2 : 1;
        // bubble the group up to dest, one window at a time
        for (;;) {
            int wb = std::max(dest, we - W);
            if (wb > 0 && sub[wb - 1] != 0.0) wb++;         // do not cut a 2x2 block
            int const w = we - wb;
            if (w < 2) break;
            // window -> host
            SN_HIP_CHECK(hipMemcpy2DAsync(ws.hT, (size_t)ldh * 8, dS + (size_t)wb * ldS + wb, (size_t)ldS * 8,
                (size_t)w * 8, w, hipMemcpyDeviceToHost, s));
            SN_HIP_CHECK(hipStreamSynchronize(s));
            for (int j = 0; j < w; j++)
                for (int i = 0; i < w; i++) ws.hZ[(size_t)j * ldh + i] = (i == j) ? 1.0 : 0.0;
            int failed = 0;
            int const placed = host::reorder_window(w, ws.hT, ldh, ws.hZ, ldh, sel.data() + wb, &failed);
            // window and factor -> device; off-diagonal updates
            SN_HIP_CHECK(hipMemcpy2DAsync(dS + (size_t)wb * ldS + wb, (size_t)ldS * 8, ws.hT, (size_t)ldh * 8,
                (size_t)w * 8, w, hipMemcpyHostToDevice, s));
            SN_HIP_CHECK(hipMemcpy2DAsync(ws.dZ, (size_t)w * 8, ws.hZ, (size_t)ldh * 8, (size_t)w * 8, w,
                hipMemcpyHostToDevice, s));
            if (wb > 0) dgemm_right_inplace(s, wb, w, ws.dZ, w, dS + (size_t)wb * ldS, ldS);
            if (n - we > 0) dgemm_left_inplace(s, w, n - we, ws.dZ, w, dS + (size_t)we * ldS + wb, ldS);
            if (dQ) dgemm_right_inplace(s, n, w, ws.dZ, w, dQ + (size_t)wb * ldQ, ldQ);
            windows += 1.0;
            flops += 2.0 * w * w * ((double)wb + (n - we) + (dQ ? n : 0));
            // the host buffers are reused by the next window
            SN_HIP_CHECK(hipStreamSynchronize(s));
            // new block structure inside the window
            for (int i = 0; i + 1 < w; i++) sub[wb + i] = ws.hT[(size_t)i * ldh + i + 1];
            if (failed) { rc = STARNEIG_PARTIAL_REORDERING; break; }
            we = wb + placed;
            if (wb == dest) break;
        }
    }
    // marks of the correctly placed eigenvalues (reorder/interface.c:166-187)
    for (int i = 0; i < n; i++) selected[i] = (i < dest && rc == STARNEIG_SUCCESS) ? 1 : 0;
    if (rc != STARNEIG_SUCCESS) {
        int placed = 0;
        while (placed < n && sel[placed]) placed++;
        for (int i = 0; i < placed; i++) selected[i] = 1;
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
    if (stats) { stats[0] = windows; stats[1] = flops; }
    return rc;
}

} // namespace sn
