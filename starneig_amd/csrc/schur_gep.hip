// Multi-shift QZ with aggressive early deflation on one MI355X: generalized Schur reduction of
// a Hessenberg-triangular pencil (A, B) -- BASELINE config 5, row S9 of SURVEY 8a.
//
// Reference: the same state machine as the standard case (schur/core.c:2226-2336) with the
// generalized branches of the window kernels (schur/cpu_utils.c:1168-1810: left reflectors
// from A's bulge columns, right reflectors that restore B's triangularity,
// create_right_reflector :998-1040).  The design mirrors schur.hip: A, B, Q and Z never leave
// HBM; one workgroup chases one chain of bulges through a 64-row diagonal window of the
// pencil held in LDS together with BOTH accumulated orthogonal factors (4 x 64 x 65 doubles =
// 133 KB of the 160 KB LDS); all off-diagonal updates are in-place fp64-MFMA GEMMs on a
// near/far stream pair.  AED windows and small blocks are reduced on the host
// (schur_host_gep.hip).  Infinite eigenvalues (push_inf_*, cpu_utils.c:360-799,
// core.c:475-552): before a block is worked on, every diagonal entry of B below the infinity
// threshold (u*||B||_F by default) is chased to the top of the block through a chain of <= 128-row
// windows on pinned host copies -- the rotations of a window reach the rest of A, B, Q, Z as
// in-place MFMA GEMMs -- and deflated there with beta = 0 exactly (GepDriver::push_infinite).
#include "common.h"
#include "schur_host.h"
#include "dgemm_tile.h"
#include "schur_common.h"
#include "tuning.h"
#include <vector>
#include <algorithm>
#include <cmath>
#include <cfloat>
#include <chrono>
#include <cstdlib>
#include <starneig/error.h>

namespace sn {

void sumsq_diff(hipStream_t s, int m, int n, double const *X, int ldx, double const *Y, int ldy,
    double ident, double *acc);
void sumsq_ordered(hipStream_t s, int m, int n, double const *X, int ldx, double *part, double *out);

constexpr int GWS = 64;             // diagonal window of the pencil held in LDS
constexpr int GNB = 10;             // bulges per chain: 6*GNB + 1 <= GWS
constexpr int GLD = GWS + 1;        // odd leading dimension: conflict-free row AND column walks
constexpr int GEP_CHASE_THREADS = 1024;
constexpr int GEP_CHASE_LDS_BYTES = (4 * GWS * GLD + 8 * GNB + 16) * 8;
constexpr int GEP_LDS_BYTES_L = GemmCfg<64, 128, 16, true, false>::LDS_BYTES;
constexpr int GEP_LDS_BYTES_R = GemmCfg<128, 64, 16, false, false>::LDS_BYTES;

// First column of (A B^-1 - s1 I)(A B^-1 - s2 I) for the leading 3x3 of a Hessenberg-
// triangular pencil (Golub & Van Loan Alg. 7.7.2 written for a shift pair; the reference
// gets the same vector from its bulge-introduction branch, cpu_utils.c:1259-1330).
__device__ __forceinline__ void gep_shift_vector(double const *A, double const *B,
    double sr1, double si1, double sr2, double si2, double *v)
{
    double const b00 = B[0], b01 = B[GLD], b11 = B[GLD + 1];
    if (b00 == 0.0 || b11 == 0.0) { v[0] = v[1] = v[2] = 0.0; return; }
    double const sum = sr1 + sr2, prod = sr1 * sr2 - si1 * si2;
    double const a00 = A[0], a10 = A[1], a01 = A[GLD], a11 = A[GLD + 1], a21 = A[GLD + 2];
    double const z0 = a00 / b00, z1 = a10 / b00;
    double const t1 = z1 / b11, t0 = (z0 - b01 * t1) / b00;
    v[0] = a00 * t0 + a01 * t1 - sum * z0 + prod;
    v[1] = a10 * t0 + a11 * t1 - sum * z1;
    v[2] = a21 * t1;
}

// One workgroup chases one chain of bulges through one diagonal window of the pencil.
// Per column step: (1) one lane per bulge builds the left reflector from A's bulge column,
// (2) all lanes apply it to the rows of A and B and accumulate it into Uq, (3) one lane per
// bulge builds the right reflector that annihilates B's fill-in (its first column is
// orthogonal to rows 2,3 of B's 3x3 block), (4) all lanes apply it to the columns of A, B
// and accumulate it into Uz.  Bulges sit 3 columns apart so the reflectors of one step
// touch disjoint rows/columns.
__global__ __launch_bounds__(GEP_CHASE_THREADS)
void gep_chase_kernel(SweepStep const step, double *__restrict__ Ag, int ldA,
    double *__restrict__ Bg, int ldB, double *__restrict__ Uout,
    double const *__restrict__ sr, double const *__restrict__ si)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *WA = lds, *WB = WA + GWS * GLD, *UQ = WB + GWS * GLD, *UZ = UQ + GWS * GLD;
    double *R = UZ + GWS * GLD;                 // per bulge {v1,v2,tau, w1,w2,tauz}
    int *Ri = reinterpret_cast<int *>(R + 6 * GNB);     // per bulge {row0, len}
    ChaseTask const t = make_task(step, blockIdx.x);
    int const n = t.n, nb = t.nb, tid = threadIdx.x;
    bool const introduce = t.flags & 1, finalize = t.flags & 2;

    for (int idx = tid; idx < n * n; idx += GEP_CHASE_THREADS) {
        int r = idx % n, c = idx / n;
        WA[c * GLD + r] = Ag[(size_t)(t.lo + c) * ldA + t.lo + r];
        WB[c * GLD + r] = Bg[(size_t)(t.lo + c) * ldB + t.lo + r];
        double const e = (r == c) ? 1.0 : 0.0;
        UQ[c * GLD + r] = e; UZ[c * GLD + r] = e;
    }
    __syncthreads();

    int const left = introduce ? 2 - 3 * nb : 0;
    int const right = finalize ? n - 2 : t.right;
    for (int begin = left; begin < right; begin++) {
        // (1) left reflectors
        if (tid < nb) {
            int const i = tid, j = begin + 3 * i;
            int len = 0;
            double beta = 0.0, v1 = 0.0, v2 = 0.0, tau = 0.0;
            if (j >= -1 && j < n - 2) {
                if (j == -1) {
                    double x[3];
                    gep_shift_vector(WA, WB, sr[t.shift_off + 2 * i], si[t.shift_off + 2 * i],
                        sr[t.shift_off + 2 * i + 1], si[t.shift_off + 2 * i + 1], x);
                    len = 3;
                    small_reflector(3, x, beta, v1, v2, tau);
                } else {
                    len = (j == n - 3) ? 2 : 3;
                    double *col = WA + j * GLD + j + 1;
                    small_reflector(len, col, beta, v1, v2, tau);
                    col[0] = beta; col[1] = 0.0;
                    if (len == 3) col[2] = 0.0;
                }
            }
            R[6 * i + 0] = v1; R[6 * i + 1] = v2; R[6 * i + 2] = tau;
            Ri[2 * i + 0] = j + 1; Ri[2 * i + 1] = len;
        }
        __syncthreads();
        // (2) left: rows row0..row0+len-1 of A and B (columns >= row0); columns of Uq
        for (int item = tid; item < nb * n * 3; item += GEP_CHASE_THREADS) {
            int const i = item / (3 * n), rr = item - i * 3 * n;
            int const len = Ri[2 * i + 1], row0 = Ri[2 * i];
            double const tau = R[6 * i + 2];
            if (len == 0 || tau == 0.0) continue;
            double const v1 = R[6 * i], v2 = R[6 * i + 1];
            if (rr < 2 * n) {
                int const c = rr < n ? rr : rr - n;
                if (c < row0) continue;
                double *p = (rr < n ? WA : WB) + c * GLD + row0;
                double x0 = p[0], x1 = p[1], x2 = (len == 3) ? p[2] : 0.0;
                double s = tau * (x0 + v1 * x1 + v2 * x2);
                p[0] = x0 - s; p[1] = x1 - s * v1;
                if (len == 3) p[2] = x2 - s * v2;
            } else {
                double *p = UQ + row0 * GLD + (rr - 2 * n);
                double x0 = p[0], x1 = p[GLD], x2 = (len == 3) ? p[2 * GLD] : 0.0;
                double s = tau * (x0 + v1 * x1 + v2 * x2);
                p[0] = x0 - s; p[GLD] = x1 - s * v1;
                if (len == 3) p[2 * GLD] = x2 - s * v2;
            }
        }
        __syncthreads();
        // (3) right reflectors: H e1 orthogonal to the rows of B that must become (0, *, *)
        if (tid < nb) {
            int const i = tid, len = Ri[2 * i + 1], row0 = Ri[2 * i];
            double beta = 0.0, w1 = 0.0, w2 = 0.0, tauz = 0.0;
            if (len == 3) {
                double const *b = WB + row0 * GLD + row0;
                double const p0 = b[1], p1 = b[GLD + 1], p2 = b[2 * GLD + 1];       // row row0+1
                double const q0 = b[2], q1 = b[GLD + 2], q2 = b[2 * GLD + 2];       // row row0+2
                // scale the rows first (by powers of two: exact): the cross product of two tiny rows
                // would underflow
                double const mp = fmax(fabs(p0), fmax(fabs(p1), fabs(p2)));
                double const mq = fmax(fabs(q0), fmax(fabs(q1), fabs(q2)));
                if (mp > 0.0 && mq > 0.0) {
                    int const ep = -ilogb(mp), eq = -ilogb(mq);
                    double const a0 = scalbn(p0, ep), a1 = scalbn(p1, ep), a2 = scalbn(p2, ep);
                    double const c0 = scalbn(q0, eq), c1 = scalbn(q1, eq), c2 = scalbn(q2, eq);
                    double x[3] = {a1 * c2 - a2 * c1, a2 * c0 - a0 * c2, a0 * c1 - a1 * c0};
                    small_reflector(3, x, beta, w1, w2, tauz);
                } else if (mp > 0.0) {
                    // row row0+2 vanished: only row row0+1 constrains H e1 -- rotate in (0,1)
                    double x[3] = {p1, -p0, 0.0};
                    small_reflector(3, x, beta, w1, w2, tauz);
                } else if (mq > 0.0) {
                    double x[3] = {q1, -q0, 0.0};
                    small_reflector(3, x, beta, w1, w2, tauz);
                }
            } else if (len == 2) {
                double const *b = WB + row0 * GLD + row0;
                double x[2] = {b[GLD + 1], -b[1]};
                small_reflector(2, x, beta, w1, w2, tauz);
            }
            R[6 * i + 3] = w1; R[6 * i + 4] = w2; R[6 * i + 5] = tauz;
        }
        __syncthreads();
        // (4) right: columns row0..row0+len-1; A rows 0..min(n-1,row0+3), B rows 0..row0+len-1,
        //     all rows of Uz
        for (int item = tid; item < nb * n * 3; item += GEP_CHASE_THREADS) {
            int const i = item / (3 * n), rr = item - i * 3 * n;
            int const len = Ri[2 * i + 1], row0 = Ri[2 * i];
            if (len == 0) continue;
            double const w1 = R[6 * i + 3], w2 = R[6 * i + 4], tauz = R[6 * i + 5];
            double *M; int r; bool clean = false;
            if (rr < n) { r = rr; if (r > row0 + 3) continue; M = WA; }
            else if (rr < 2 * n) { r = rr - n; if (r > row0 + len - 1) continue; M = WB; clean = r > row0; }
            else { r = rr - 2 * n; M = UZ; }
            double *p = M + row0 * GLD + r;
            if (tauz != 0.0) {
                double x0 = p[0], x1 = p[GLD], x2 = (len == 3) ? p[2 * GLD] : 0.0;
                double s = tauz * (x0 + w1 * x1 + w2 * x2);
                p[0] = x0 - s; p[GLD] = x1 - s * w1;
                if (len == 3) p[2 * GLD] = x2 - s * w2;
            }
            if (clean) p[0] = 0.0;
        }
        __syncthreads();
    }

    double *Uo = Uout + (size_t)blockIdx.x * 2 * GWS * GWS;
    for (int idx = tid; idx < n * n; idx += GEP_CHASE_THREADS) {
        int r = idx % n, c = idx / n;
        Ag[(size_t)(t.lo + c) * ldA + t.lo + r] = WA[c * GLD + r];
        Bg[(size_t)(t.lo + c) * ldB + t.lo + r] = WB[c * GLD + r];
        Uo[c * GWS + r] = UQ[c * GLD + r];
        Uo[GWS * GWS + c * GWS + r] = UZ[c * GLD + r];
    }
}

// Off-diagonal updates of all chains of one step, A and B in one launch.
//   MODE 2 ("near"): X(win, next `adv` columns) <- Uq^T .      X in {A, B}
//   MODE 0 ("far") : the remaining columns right of the window
//   MODE 1         : X(above win, win) <- . Uz  (X in {A, B})
//   MODE 3         : Q(:, win) <- . Uq,  Z(:, win) <- . Uz   (lazy stream)
template <int MODE>
__global__ __launch_bounds__(256, 2)
void gep_update_kernel(SweepStep const step, double *__restrict__ A, int ldA,
    double *__restrict__ B, int ldB, double *__restrict__ Q, int ldQ,
    double *__restrict__ Z, int ldZ, int n, double const *__restrict__ U)
{
    int const k = blockIdx.y % step.ntasks, which = blockIdx.y / step.ntasks;
    ChaseTask const t = make_task(step, k);
    double const *Uq = U + (size_t)k * 2 * GWS * GWS, *Uz = Uq + GWS * GWS;
    int const w = t.n, lo = t.lo;
    if (MODE == 0 || MODE == 2) {
        int const c0 = (MODE == 2) ? lo + w : lo + w + step.adv;
        int ncols = n - c0;
        if (MODE == 2) ncols = min(ncols, step.adv);
        if ((int)blockIdx.x * 128 >= ncols) return;
        int const ld = which ? ldB : ldA;
        double *X = (which ? B : A) + (size_t)c0 * ld + lo;
        gemm_tile<64, 128, 16, true, false>(w, ncols, w, 1.0, Uq, GWS, X, ld, 0.0, X, ld, 0, blockIdx.x);
    } else if (MODE == 1) {
        // which: 0 = A, 1 = B
        double *X = which ? B : A;
        int const ld = which ? ldB : ldA, rows = lo;
        if ((int)blockIdx.x * 128 >= rows) return;
        X += (size_t)lo * ld;
        gemm_tile<128, 64, 16, false, false>(rows, w, w, 1.0, X, ld, Uz, GWS, 0.0, X, ld, blockIdx.x, 0);
    } else {
        // MODE 3 -- which: 0 = Z (or Q when there is no Z), 1 = Q.  Nothing on the GPU reads
        // Q or Z: these run on the lazy stream (see schur.hip)
        double *X; int ld; double const *Uk = Uz;
        if (which == 0 && Z) { X = Z; ld = ldZ; }
        else { X = Q; ld = ldQ; Uk = Uq; }
        if (X == nullptr || (int)blockIdx.x * 128 >= n) return;
        X += (size_t)lo * ld;
        gemm_tile<128, 64, 16, false, false>(n, w, w, 1.0, X, ld, Uk, GWS, 0.0, X, ld, blockIdx.x, 0);
    }
}

// sub[i] = A(i+1,i) for i in [0,hi-1); entries below the threshold become exact zeros
// bflag[i] = 1 where |B(i,i)| is below the infinity threshold (schur/core.c:475-552 scans the
// same way before it inserts its push_inf tasks)
__global__ void gep_scan_subdiag_kernel(int hi, double *__restrict__ A, int ldA, double thres,
    double *__restrict__ sub, double const *__restrict__ B, int ldB, double thres_inf,
    double *__restrict__ bflag)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < hi) bflag[i] = (fabs(B[(size_t)i * ldB + i]) < thres_inf) ? 1.0 : 0.0;
    if (i >= hi - 1) return;
    double *p = A + (size_t)i * ldA + i + 1;
    double v = *p;
    if (v != 0.0) {
        bool small;
        if (thres > 0.0) small = fabs(v) < thres;
        else small = fabs(v) <= DBL_EPSILON * (fabs(A[(size_t)i * ldA + i]) + fabs(p[ldA]));
        if (small) { v = 0.0; *p = 0.0; }
    }
    sub[i] = v;
}

__global__ void gep_set_entry_kernel(double *p, double v) { *p = v; }

// ---- workspace --------------------------------------------------------------------------
struct GepWorkspace {
    int n = 0, nwmax = 0, max_chains = 0;
    double *dU = nullptr;           // EV_RING x max_chains x {Uq, Uz} x GWS x GWS
    double *dShiftR = nullptr, *dShiftI = nullptr, *dSub = nullptr;
    double *dQl = nullptr, *dZl = nullptr, *dTmp = nullptr, *dAcc = nullptr;
    double *hA = nullptr, *hB = nullptr, *hQ = nullptr, *hZ = nullptr, *hSub = nullptr;   // pinned
    bool attr_set = false;
    hipStream_t far = nullptr, qs = nullptr;    // far updates of A, B; lazy updates of Q, Z
    static constexpr int EV_RING = 2048;        // largest ring of per-step factor buffers
    int ring = EV_RING;                         // ring in use
    hipEvent_t near_done[EV_RING] = {}, far_done[EV_RING] = {};
    static constexpr int FLUSH_RING = 16;
    hipEvent_t q_done[FLUSH_RING] = {};
    long issued_total = 0, flush_total = 0;
    std::vector<long> slot_flush = std::vector<long>(EV_RING, -1);
    static constexpr int Z_RING = 64;           // AED / small-block factors waiting for the lazy stream (at most)
    int z_ring = Z_RING;                        // slots in use: fewer for very large AED windows (2 GB of factors at most)
    double *dQZq = nullptr, *dTmpQ = nullptr;   // z_ring x {Ql, Zl}; scratch of the lazy stream
    hipEvent_t z_ready[Z_RING] = {}, z_done[Z_RING] = {};
    long z_total = 0;
    hipEvent_t lazy_mark = nullptr;

    void release() {
        void **dptrs[] = {(void **)&dU, (void **)&dShiftR, (void **)&dShiftI, (void **)&dSub,
            (void **)&dQl, (void **)&dZl, (void **)&dTmp, (void **)&dAcc, (void **)&dQZq, (void **)&dTmpQ};
        for (auto p : dptrs) if (*p) { SN_HIP_CHECK(hipFree(*p)); *p = nullptr; }
        void **hptrs[] = {(void **)&hA, (void **)&hB, (void **)&hQ, (void **)&hZ, (void **)&hSub};
        for (auto p : hptrs) if (*p) { SN_HIP_CHECK(hipHostFree(*p)); *p = nullptr; }
        n = nwmax = max_chains = 0;
    }
    void ensure(int n_, int nw_, int chains_) {
        if (n_ <= n && nw_ <= nwmax && chains_ <= max_chains) return;
        release();
        n = n_; nwmax = nw_; max_chains = chains_;
        size_t const w2 = (size_t)(nwmax + 24) * nwmax * 8;
        ring = std::min(EV_RING, std::max(64, 4 * (n / 30 + 128)));
        std::fill(slot_flush.begin(), slot_flush.end(), -1L);
        SN_HIP_CHECK(hipMalloc((void **)&dU, (size_t)ring * max_chains * 2 * GWS * GWS * 8));
        z_ring = (int)std::max<size_t>(2, std::min<size_t>(Z_RING, ((size_t)2 << 30) / ((size_t)2 * nwmax * nwmax * 8)));
        z_total = 0;
        SN_HIP_CHECK(hipMalloc((void **)&dQZq, (size_t)z_ring * 2 * nwmax * nwmax * 8));
        SN_HIP_CHECK(hipMalloc((void **)&dTmpQ, (size_t)n * nwmax * 8));
        SN_HIP_CHECK(hipMalloc((void **)&dShiftR, (size_t)8 * nwmax * 8));
        SN_HIP_CHECK(hipMalloc((void **)&dShiftI, (size_t)8 * nwmax * 8));
        SN_HIP_CHECK(hipMalloc((void **)&dSub, (size_t)2 * n * 8));
        SN_HIP_CHECK(hipMalloc((void **)&dQl, w2));
        SN_HIP_CHECK(hipMalloc((void **)&dZl, w2));
        SN_HIP_CHECK(hipMalloc((void **)&dTmp, (size_t)n * nwmax * 8));
        SN_HIP_CHECK(hipMalloc((void **)&dAcc, 4 * 8));
        SN_HIP_CHECK(hipHostMalloc((void **)&hA, w2, hipHostMallocDefault));
        SN_HIP_CHECK(hipHostMalloc((void **)&hB, w2, hipHostMallocDefault));
        SN_HIP_CHECK(hipHostMalloc((void **)&hQ, w2, hipHostMallocDefault));
        SN_HIP_CHECK(hipHostMalloc((void **)&hZ, w2, hipHostMallocDefault));
        SN_HIP_CHECK(hipHostMalloc((void **)&hSub, (size_t)2 * n * 8, hipHostMallocDefault));
        if (!attr_set) {
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)gep_chase_kernel,
                hipFuncAttributeMaxDynamicSharedMemorySize, GEP_CHASE_LDS_BYTES));
            int lo_prio = 0, hi_prio = 0;
            SN_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio));
            make_stream(&far, true, hi_prio);   // see schur.hip

            make_stream(&qs, false, lo_prio);
            SN_HIP_CHECK(hipEventCreateWithFlags(&lazy_mark, hipEventDisableTiming));
            for (int k = 0; k < FLUSH_RING; k++) SN_HIP_CHECK(hipEventCreateWithFlags(&q_done[k], hipEventDisableTiming));
            for (int k = 0; k < Z_RING; k++) {
                SN_HIP_CHECK(hipEventCreateWithFlags(&z_ready[k], hipEventDisableTiming));
                SN_HIP_CHECK(hipEventCreateWithFlags(&z_done[k], hipEventDisableTiming));
            }
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)gep_update_kernel<3>,
                hipFuncAttributeMaxDynamicSharedMemorySize, GEP_LDS_BYTES_R));
            for (int k = 0; k < EV_RING; k++) {
                SN_HIP_CHECK(hipEventCreateWithFlags(&near_done[k], hipEventDisableTiming));
                SN_HIP_CHECK(hipEventCreateWithFlags(&far_done[k], hipEventDisableTiming));
            }
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)gep_update_kernel<2>,
                hipFuncAttributeMaxDynamicSharedMemorySize, GEP_LDS_BYTES_L));
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)gep_update_kernel<0>,
                hipFuncAttributeMaxDynamicSharedMemorySize, GEP_LDS_BYTES_L));
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)gep_update_kernel<1>,
                hipFuncAttributeMaxDynamicSharedMemorySize, GEP_LDS_BYTES_R));
            attr_set = true;
        }
    }
};
// level 0: the caller's pencil; level 1: the private AED window of a blocked AED (GepDriver::large_aed)
static GepWorkspace g_gws[2];

// private matrices of a blocked AED: the window pencil, its accumulated factors, the padded pencil whose
// Hessenberg-triangular reduction restores the form of the undeflated part, local window factors
struct GepLargeBuffers {
    int cap = 0, ld = 0;
    double *dA = nullptr, *dB = nullptr, *dQ = nullptr, *dZ = nullptr;
    double *dPA = nullptr, *dPB = nullptr, *dPQ = nullptr, *dPZ = nullptr, *dTmp = nullptr;
    double *dQl = nullptr, *dZl = nullptr;
    void release() {
        double **ptrs[] = {&dA, &dB, &dQ, &dZ, &dPA, &dPB, &dPQ, &dPZ, &dTmp, &dQl, &dZl};
        for (auto p : ptrs) if (*p) { SN_HIP_CHECK(hipFree(*p)); *p = nullptr; }
        cap = 0;
    }
    void ensure(int nw) {
        if (nw <= cap) return;
        release();
        cap = nw; ld = (int)roundup((size_t)nw + 1, 16);
        size_t const bytes = (size_t)ld * (nw + 1) * 8;
        double **ptrs[] = {&dA, &dB, &dQ, &dZ, &dPA, &dPB, &dPQ, &dPZ, &dTmp};
        for (auto p : ptrs) SN_HIP_CHECK(hipMalloc((void **)p, bytes));
        SN_HIP_CHECK(hipMalloc((void **)&dQl, (size_t)128 * 128 * 8));
        SN_HIP_CHECK(hipMalloc((void **)&dZl, (size_t)128 * 128 * 8));
    }
};
static GepLargeBuffers g_glarge;
void gep_schur_release_workspace() { g_gws[0].release(); g_gws[1].release(); g_glarge.release(); }

namespace {

static inline double wall()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct GepDriver {
    hipStream_t s;
    int n; double *A; int ldA; double *B; int ldB; double *Q; int ldQ; double *Z; int ldZ;
    GepWorkspace &ws;
    SchurStats st;
    double thres_b = 0.0;       // B-side threshold of the host window kernels (0: LAPACK's BTOL per window)

    // X <- op applied in place through the scratch panel for windows wider than one GEMM tile
    void left_update(double *X, int ld, int lo, int w, double const *dU, int ldu)
    {
        int const cols = n - (lo + w);
        if (cols <= 0) return;
        double *P = X + (size_t)(lo + w) * ld + lo;
        if (w <= 128) dgemm_left_inplace(s, w, cols, dU, ldu, P, ld);
        else {
            dgemm(s, 'T', 'N', w, cols, w, 1.0, dU, ldu, P, ld, 0.0, ws.dTmp, w);
            copy_matrix(s, w, cols, ws.dTmp, w, P, ld);
        }
    }
    void right_update(double *X, int ld, int rows, int lo, int w, double const *dU, int ldu)
    {
        if (rows <= 0) return;
        double *P = X + (size_t)lo * ld;
        if (w <= 128) dgemm_right_inplace(s, rows, w, dU, ldu, P, ld);
        else {
            dgemm(s, 'N', 'N', rows, w, w, 1.0, P, ld, dU, ldu, 0.0, ws.dTmp, rows);
            copy_matrix(s, rows, w, ws.dTmp, rows, P, ld);
        }
    }
    // (A,B)(win, right) <- Ql^T .,  (A,B)(above, win) <- . Zl,  Q(:,win) <- . Ql,  Z(:,win) <- . Zl.
    // Q and Z are updated on the lazy stream from private copies of the factors.
    void apply_transform(int lo, int w, double const *dQl, double const *dZl, int ldu)
    {
        left_update(A, ldA, lo, w, dQl, ldu);
        left_update(B, ldB, lo, w, dQl, ldu);
        right_update(A, ldA, lo, lo, w, dZl, ldu);
        right_update(B, ldB, lo, lo, w, dZl, ldu);
        if (Q || Z) {
            int const slot = (int)(ws.z_total % ws.z_ring);
            double *Qc = ws.dQZq + (size_t)slot * 2 * ws.nwmax * ws.nwmax, *Zc = Qc + (size_t)ws.nwmax * ws.nwmax;
            if (ws.z_total >= ws.z_ring) SN_HIP_CHECK(hipStreamWaitEvent(s, ws.z_done[slot], 0));
            copy_matrix(s, w, w, dQl, ldu, Qc, w);
            copy_matrix(s, w, w, dZl, ldu, Zc, w);
            SN_HIP_CHECK(hipEventRecord(ws.z_ready[slot], s));
            SN_HIP_CHECK(hipStreamWaitEvent(ws.qs, ws.z_ready[slot], 0));
            auto lazy_right = [&](double *X, int ld, double const *U) {
                double *P = X + (size_t)lo * ld;
                if (w <= 128) dgemm_right_inplace(ws.qs, n, w, U, w, P, ld);
                else {
                    dgemm(ws.qs, 'N', 'N', n, w, w, 1.0, P, ld, U, w, 0.0, ws.dTmpQ, n);
                    copy_matrix(ws.qs, n, w, ws.dTmpQ, n, P, ld);
                }
            };
            if (Q) lazy_right(Q, ldQ, Qc);
            if (Z) lazy_right(Z, ldZ, Zc);
            SN_HIP_CHECK(hipEventRecord(ws.z_done[slot], ws.qs));
            ws.z_total++;
        }
        st.gemm_flops += 2.0 * w * w * (2.0 * (n - lo - w) + 2.0 * lo + (Q ? n : 0) + (Z ? n : 0));
    }

    // padded host leading dimension (see schur.hip)
    static int host_ld(int w) { int ld = (w + 8 + 7) / 8 * 8; if ((ld / 8) % 2 == 0) ld += 8; return ld; }    // an odd number of cache lines (schur.hip)
    void download_windows(int lo, int w)
    {
        double t0 = wall();
        size_t const hp = (size_t)host_ld(w) * 8;
        SN_HIP_CHECK(hipMemcpy2DAsync(ws.hA, hp, A + (size_t)lo * ldA + lo, (size_t)ldA * 8,
            (size_t)w * 8, w, hipMemcpyDeviceToHost, s));
        SN_HIP_CHECK(hipMemcpy2DAsync(ws.hB, hp, B + (size_t)lo * ldB + lo, (size_t)ldB * 8,
            (size_t)w * 8, w, hipMemcpyDeviceToHost, s));
        SN_HIP_CHECK(hipStreamSynchronize(s));
        st.wait_s += wall() - t0;
    }
    void upload_windows(int lo, int w)
    {
        size_t const hp = (size_t)host_ld(w) * 8;
        SN_HIP_CHECK(hipMemcpy2DAsync(A + (size_t)lo * ldA + lo, (size_t)ldA * 8, ws.hA, hp,
            (size_t)w * 8, w, hipMemcpyHostToDevice, s));
        SN_HIP_CHECK(hipMemcpy2DAsync(B + (size_t)lo * ldB + lo, (size_t)ldB * 8, ws.hB, hp,
            (size_t)w * 8, w, hipMemcpyHostToDevice, s));
        SN_HIP_CHECK(hipMemcpy2DAsync(ws.dQl, (size_t)w * 8, ws.hQ, hp, (size_t)w * 8, w, hipMemcpyHostToDevice, s));
        SN_HIP_CHECK(hipMemcpy2DAsync(ws.dZl, (size_t)w * 8, ws.hZ, hp, (size_t)w * 8, w, hipMemcpyHostToDevice, s));
    }

    // small pencil on a host copy (row S6, GEP branch: schur/cpu_utils.c:3185-3371)
    int small_block(int lo, int w, double *real, double *imag, double *beta)
    {
        int const ldh = host_ld(w);
        download_windows(lo, w);
        for (int j = 0; j < w; j++)
            for (int i = 0; i < w; i++) ws.hQ[(size_t)j * ldh + i] = ws.hZ[(size_t)j * ldh + i] = (i == j) ? 1.0 : 0.0;
        std::vector<double> ar(w), ai(w), be(w);
        int info = host::gep_small_schur(w, ws.hA, ldh, ws.hB, ldh, ws.hQ, ldh, ws.hZ, ldh, w, ar.data(), ai.data(), be.data(), thres_b);
        if (info != 0) return info;
        upload_windows(lo, w);
        apply_transform(lo, w, ws.dQl, ws.dZl, w);
        SN_HIP_CHECK(hipStreamSynchronize(s));
        if (real) for (int i = 0; i < w; i++) { real[lo + i] = ar[i]; imag[lo + i] = ai[i]; beta[lo + i] = be[i]; }
        st.small_solves++;
        return 0;
    }

    // Infinite eigenvalues of the active block [ilo, ihi): every diagonal entry of B below the
    // threshold is chased to the top of the block -- window by window on pinned host copies
    // (host::gep_push_inf_window), the rows and columns outside a window see the accumulated
    // rotations through the usual GEMM updates -- and deflated there: A(to+1,to) = 0, B(to,to) = 0
    // exactly, eigenvalue (A(to,to), 0).  Returns the number of deflated eigenvalues.
    double prof_inf_s = 0.0; int prof_inf_windows = 0;
    int push_infinite(int ilo, int ihi, double thres_inf, double *real, double *imag, double *beta)
    {
        double const t_begin = wall();
        int const W = std::min(128, ws.nwmax);
        int lo = ilo, hi = ihi, count = 0;      // the block shrinks from the end an eigenvalue leaves through
        std::vector<double> dg(ihi - ilo);
        auto identity = [&](int w, int ldh) {
            for (int j = 0; j < w; j++)
                for (int i = 0; i < w; i++) ws.hQ[(size_t)j * ldh + i] = ws.hZ[(size_t)j * ldh + i] = (i == j) ? 1.0 : 0.0;
        };
        while (lo + 1 < hi) {
            int const len = hi - lo;
            SN_HIP_CHECK(hipMemcpy2DAsync(dg.data(), 8, B + (size_t)lo * ldB + lo, (size_t)(ldB + 1) * 8, 8, len,
                hipMemcpyDeviceToHost, s));
            SN_HIP_CHECK(hipStreamSynchronize(s));
            int z = -1;
            for (int i = 0; i < len; i++) if (std::fabs(dg[i]) < thres_inf) { z = lo + i; break; }
            if (z < 0) break;
            double alpha = 0.0;
            int cur = z;
            // towards the nearer end of the block (the zeros that the iteration itself produces sit
            // in the AED region at the bottom: one window; the reference pushes to the top only)
            if (z - lo <= hi - 1 - z) {
                for (;;) {
                    int const we = std::min(hi, cur + 2), wb = std::max(lo, we - W), w = we - wb;
                    int const ldh = host_ld(w);
                    download_windows(wb, w);
                    identity(w, ldh);
                    bool const last = (wb == lo);
                    host::gep_push_inf_window(w, ws.hA, ldh, ws.hB, ldh, ws.hQ, ldh, ws.hZ, ldh, cur - wb, 0, last ? 1 : 0);
                    if (last) alpha = ws.hA[0];
                    upload_windows(wb, w);
                    apply_transform(wb, w, ws.dQl, ws.dZl, w);
                    SN_HIP_CHECK(hipStreamSynchronize(s));
                    cur = wb; prof_inf_windows++;
                    if (last) break;
                }
                if (real) { real[lo] = alpha; imag[lo] = 0.0; beta[lo] = 0.0; }
                lo++;
            } else {
                for (;;) {
                    int const wb = std::max(lo, cur - 1), we = std::min(hi, wb + W), w = we - wb;
                    int const ldh = host_ld(w);
                    download_windows(wb, w);
                    identity(w, ldh);
                    bool const last = (we == hi);
                    host::gep_push_inf_down_window(w, ws.hA, ldh, ws.hB, ldh, ws.hQ, ldh, ws.hZ, ldh, cur - wb, last ? 1 : 0);
                    if (last) alpha = ws.hA[(size_t)(w - 1) * ldh + (w - 1)];
                    upload_windows(wb, w);
                    apply_transform(wb, w, ws.dQl, ws.dZl, w);
                    SN_HIP_CHECK(hipStreamSynchronize(s));
                    cur = we - 1; prof_inf_windows++;
                    if (last) break;
                }
                if (real) { real[hi - 1] = alpha; imag[hi - 1] = 0.0; beta[hi - 1] = 0.0; }
                hi--;
            }
            count++;
        }
        st.inf_deflated += count;
        prof_inf_s += wall() - t_begin;
        return count;
    }

    // ---- blocked AED for windows above aed_parallel_hard_limit (row S5 on the pencil side; reference
    // schur/core.c:1423-1551 perform_large_aed serves both problems, :1070-1252 perform_deflate_step, :783-1052
    // perform_deflate_finalize).  The generalized twin of Driver::large_aed in schur.hip: the window pencil is
    // copied into private matrices and reduced to generalized Schur form RECURSIVELY by this same device path
    // (level 1: chase kernels, MFMA updates, small host AEDs); the deflation checks run over <= 96-row diagonal
    // windows from the bottom up (host::gep_deflate_window on pinned copies, the rest of the private pencil and
    // of its factors sees the swaps as in-place MFMA GEMMs); undeflatable blocks are carried along and flushed
    // to the top of the AED window in batches (host::gep_reorder_window chains); the spike is embedded as the
    // first column of a padded pencil whose Hessenberg-triangular reduction (hessenberg_triangular_device)
    // restores the form of the undeflated part.  Nothing outside the private matrices is touched before the
    // outcome is known.
    void large_window_updates(GepLargeBuffers &L, int nw, int wb, int w)
    {
        int const ld = L.ld, rc = nw - (wb + w);
        if (wb > 0) {
            dgemm_right_inplace(s, wb, w, L.dZl, w, L.dA + (size_t)wb * ld, ld);
            dgemm_right_inplace(s, wb, w, L.dZl, w, L.dB + (size_t)wb * ld, ld);
        }
        if (rc > 0) {
            dgemm_left_inplace(s, w, rc, L.dQl, w, L.dA + (size_t)(wb + w) * ld + wb, ld);
            dgemm_left_inplace(s, w, rc, L.dQl, w, L.dB + (size_t)(wb + w) * ld + wb, ld);
        }
        dgemm_right_inplace(s, nw, w, L.dQl, w, L.dQ + (size_t)wb * ld, ld);
        dgemm_right_inplace(s, nw, w, L.dZl, w, L.dZ + (size_t)wb * ld, ld);
        st.gemm_flops += 2.0 * w * w * (2.0 * wb + 2.0 * rc + 2.0 * nw);
    }
    void large_window_to_host(GepLargeBuffers &L, int wb, int w, int ldh)
    {
        size_t const hp = (size_t)ldh * 8;
        SN_HIP_CHECK(hipMemcpy2DAsync(ws.hA, hp, L.dA + (size_t)wb * L.ld + wb, (size_t)L.ld * 8, (size_t)w * 8, w, hipMemcpyDeviceToHost, s));
        SN_HIP_CHECK(hipMemcpy2DAsync(ws.hB, hp, L.dB + (size_t)wb * L.ld + wb, (size_t)L.ld * 8, (size_t)w * 8, w, hipMemcpyDeviceToHost, s));
        SN_HIP_CHECK(hipStreamSynchronize(s));
        for (int j = 0; j < w; j++)
            for (int i = 0; i < w; i++) ws.hQ[(size_t)j * ldh + i] = ws.hZ[(size_t)j * ldh + i] = (i == j) ? 1.0 : 0.0;
    }
    void large_window_to_device(GepLargeBuffers &L, int wb, int w, int ldh)
    {
        size_t const hp = (size_t)ldh * 8;
        SN_HIP_CHECK(hipMemcpy2DAsync(L.dA + (size_t)wb * L.ld + wb, (size_t)L.ld * 8, ws.hA, hp, (size_t)w * 8, w, hipMemcpyHostToDevice, s));
        SN_HIP_CHECK(hipMemcpy2DAsync(L.dB + (size_t)wb * L.ld + wb, (size_t)L.ld * 8, ws.hB, hp, (size_t)w * 8, w, hipMemcpyHostToDevice, s));
        SN_HIP_CHECK(hipMemcpy2DAsync(L.dQl, (size_t)w * 8, ws.hQ, hp, (size_t)w * 8, w, hipMemcpyHostToDevice, s));
        SN_HIP_CHECK(hipMemcpy2DAsync(L.dZl, (size_t)w * 8, ws.hZ, hp, (size_t)w * 8, w, hipMemcpyHostToDevice, s));
    }

    double prof_laed[4] = {0, 0, 0, 0}; int prof_laed_calls = 0, prof_laed_windows = 0;
    host::AedResult large_aed(int kw, int nw, double sub, double thres, double thres_inf, SchurParams const &prm,
        double *spike, double *sr, double *si)
    {
        host::AedResult res{0, 0, 0};
        double const tl0 = wall(); prof_laed_calls++;
        GepLargeBuffers &L = g_glarge;
        L.ensure(nw);
        int const ld = L.ld;
        constexpr int WD = 128;                 // reorder window (in-place update tiles)
        constexpr int WDD = 96;                 // deflation window: its undeflatable blocks must fit a reorder window with room to move
        copy_matrix(s, nw, nw, A + (size_t)kw * ldA + kw, ldA, L.dA, ld);
        copy_matrix(s, nw, nw, B + (size_t)kw * ldB + kw, ldB, L.dB, ld);
        set_matrix(s, nw, nw, 0.0, 1.0, L.dQ, ld);
        set_matrix(s, nw, nw, 0.0, 1.0, L.dZ, ld);
        // (1) generalized Schur form of the window, recursively on the device (default small AED windows)
        std::vector<double> war(nw), wai(nw), wbe(nw);
        // The sub-problem INHERITS the thresholds the parent resolved from the whole pencil (the reference builds it
        // with starneig_build_process_args_from, schur/core.c:1525 / :2426-2460): positive values are taken as given
        // by the recursive call, so u ||A||_F, u ||B||_F are not recomputed from the window (they would come out
        // smaller than the `thres` the deflation checks below apply to the same window).  0 stands for the LAPACK
        // criteria, which are per entry / per window by definition.
        SchurParams p1;
        p1.threshold = thres > 0.0 ? thres : -3.0;
        p1.threshold_inf = thres_inf;
        p1.threshold_b = thres_b > 0.0 ? thres_b : prm.threshold_b;
        p1.host_threads = prm.host_threads;
        int const rc1 = gep_schur_device(s, nw, L.dA, ld, L.dB, ld, L.dQ, ld, L.dZ, ld, war.data(), wai.data(), wbe.data(),
            p1, nullptr, 1);
        SN_HIP_CHECK(hipStreamSynchronize(s));
        double const tl1 = wall(); prof_laed[0] += tl1 - tl0;
        if (rc1 != STARNEIG_SUCCESS) { res.failed = 1; return res; }
        // (2) the spike sub * Q(0,:) and the block structure
        std::vector<double> sp(nw), asub(nw, 0.0);
        SN_HIP_CHECK(hipMemcpy2DAsync(sp.data(), 8, L.dQ, (size_t)ld * 8, 8, nw, hipMemcpyDeviceToHost, s));
        if (nw > 1)
            SN_HIP_CHECK(hipMemcpy2DAsync(asub.data(), 8, L.dA + 1, (size_t)(ld + 1) * 8, 8, nw - 1, hipMemcpyDeviceToHost, s));
        SN_HIP_CHECK(hipStreamSynchronize(s));
        for (int j = 0; j < nw; j++) sp[j] *= sub;
        // (3) deflation windows, bottom up.  [0, top) final undeflatable, [top, bottom - carried) unchecked,
        // [bottom - carried, bottom) carried undeflatable, [bottom, nw) deflated
        int top = 0, bottom = nw, carried = 0;
        while (top < bottom - carried) {
            int const we = bottom;
            int wb = std::max(top, we - WDD);
            if (wb > top && asub[wb - 1] != 0.0) wb++;          // do not cut a 2x2 block
            int const w = we - wb, ldh = host_ld(w);
            large_window_to_host(L, wb, w, ldh); prof_laed_windows++;
            int und = 0;
            int const rej = host::gep_deflate_window(w, ws.hA, ldh, ws.hB, ldh, ws.hQ, ldh, ws.hZ, ldh, sp.data() + wb,
                sub, thres, carried, &und);
            large_window_to_device(L, wb, w, ldh);
            large_window_updates(L, nw, wb, w);
            SN_HIP_CHECK(hipStreamSynchronize(s));
            for (int i = 0; i + 1 < w; i++) asub[wb + i] = ws.hA[(size_t)i * ldh + i + 1];
            bottom = wb + und; carried = und;
            if (rej) { top = bottom; carried = 0; break; }      // swap rejected: stop testing
            if (wb == top) { top = bottom; carried = 0; break; }
            if (carried >= WDD / 2 || bottom - carried - top < 2) {
                // flush the carried blocks to the top of the AED window (reorder chain)
                int ge = bottom;                                // group = [ge - carried, ge)
                std::vector<int> marks(WD);
                while (ge - carried > top) {
                    int rb = std::max(top, ge - WD);
                    if (rb > top && asub[rb - 1] != 0.0) rb++;
                    int const rw = ge - rb, rldh = host_ld(rw);
                    large_window_to_host(L, rb, rw, rldh);
                    for (int i = 0; i < rw; i++) marks[i] = (i >= rw - carried) ? 1 : 0;
                    int failed = 0;
                    int const placed = host::gep_reorder_window(rw, ws.hA, rldh, ws.hB, rldh, ws.hQ, rldh, ws.hZ, rldh,
                        marks.data(), &failed);
                    {   // spike segment <- spike * Ql
                        std::vector<double> t(rw);
                        for (int j = 0; j < rw; j++) { double v = 0.0; for (int k = 0; k < rw; k++) v += sp[rb + k] * ws.hQ[(size_t)j * rldh + k]; t[j] = v; }
                        for (int j = 0; j < rw; j++) sp[rb + j] = t[j];
                    }
                    large_window_to_device(L, rb, rw, rldh);
                    large_window_updates(L, nw, rb, rw);
                    SN_HIP_CHECK(hipStreamSynchronize(s));
                    for (int i = 0; i + 1 < rw; i++) asub[rb + i] = ws.hA[(size_t)i * rldh + i + 1];
                    if (failed || placed != carried) { top = bottom; carried = 0; ge = top; break; }
                    ge = rb + carried;
                }
                if (carried > 0) { top += carried; carried = 0; }
            }
        }
        if (carried > 0) { top = bottom; carried = 0; }
        int const ns = top, nd = nw - ns;
        double const tl2 = wall(); prof_laed[1] += tl2 - tl1;
        // the window pencil on the host: shifts now, the deflated diagonal blocks for the caller later
        int const ldw = host_ld(nw);
        auto window_pencil_to_host = [&]() {
            SN_HIP_CHECK(hipMemcpy2DAsync(ws.hA, (size_t)ldw * 8, L.dA, (size_t)ld * 8, (size_t)nw * 8, nw, hipMemcpyDeviceToHost, s));
            SN_HIP_CHECK(hipMemcpy2DAsync(ws.hB, (size_t)ldw * 8, L.dB, (size_t)ld * 8, (size_t)nw * 8, nw, hipMemcpyDeviceToHost, s));
            SN_HIP_CHECK(hipStreamSynchronize(s));
        };
        window_pencil_to_host();
        res.shifts = host::gep_window_shifts(ns >= 2 ? ns : nw, ws.hA, ldw, ws.hB, ldw, sr, si);
        res.deflated = nd;
        if (nd == 0) return res;                // the caller's pencil was never touched
        // (4) spike + Hessenberg-triangular form of the undeflated part
        for (int j = 0; j < nw; j++) spike[j] = (j < ns) ? sp[j] : 0.0;
        if (ns > 1 && sub != 0.0) {
            int const np = ns + 1;
            set_matrix(s, np, np, 0.0, 0.0, L.dPA, ld);
            set_matrix(s, np, np, 0.0, 1.0, L.dPB, ld);
            set_matrix(s, np, np, 0.0, 1.0, L.dPQ, ld);
            set_matrix(s, np, np, 0.0, 1.0, L.dPZ, ld);
            copy_matrix(s, ns, ns, L.dA, ld, L.dPA + ld + 1, ld);
            copy_matrix(s, ns, ns, L.dB, ld, L.dPB + ld + 1, ld);
            SN_HIP_CHECK(hipMemcpyAsync(L.dPA + 1, sp.data(), (size_t)ns * 8, hipMemcpyHostToDevice, s));
            SN_HIP_CHECK(hipStreamSynchronize(s));      // sp is pageable
            int const hrc = hessenberg_triangular_device(s, np, L.dPA, ld, L.dPB, ld, L.dPQ, ld, L.dPZ, ld, nullptr);
            if (hrc != 0) { res.failed = 1; res.deflated = 0; return res; }
            // Uq = PQ(1:,1:), Uz = PZ(1:,1:):  (A,B)(0:ns, ns:nw) <- Uq^T . ;  Q(:,0:ns) <- . Uq ;  Z(:,0:ns) <- . Uz
            double const *Uq = L.dPQ + ld + 1, *Uz = L.dPZ + ld + 1;
            for (double *M : {L.dA, L.dB}) {
                dgemm(s, 'T', 'N', ns, nd, ns, 1.0, Uq, ld, M + (size_t)ns * ld, ld, 0.0, L.dTmp, ld);
                copy_matrix(s, ns, nd, L.dTmp, ld, M + (size_t)ns * ld, ld);
            }
            dgemm(s, 'N', 'N', nw, ns, ns, 1.0, L.dQ, ld, Uq, ld, 0.0, L.dTmp, ld);
            copy_matrix(s, nw, ns, L.dTmp, ld, L.dQ, ld);
            dgemm(s, 'N', 'N', nw, ns, ns, 1.0, L.dZ, ld, Uz, ld, 0.0, L.dTmp, ld);
            copy_matrix(s, nw, ns, L.dTmp, ld, L.dZ, ld);
            copy_matrix(s, ns, ns, L.dPA + ld + 1, ld, L.dA, ld);
            copy_matrix(s, ns, ns, L.dPB + ld + 1, ld, L.dB, ld);
            SN_HIP_CHECK(hipMemcpyAsync(spike, L.dPA + 1, 8, hipMemcpyDeviceToHost, s));
            SN_HIP_CHECK(hipStreamSynchronize(s));
            for (int r = 1; r < ns; r++) spike[r] = 0.0;
            st.gemm_flops += 2.0 * ns * ns * (2.0 * nd + 2.0 * nw);
        }
        double const tl3 = wall(); prof_laed[2] += tl3 - tl2;
        // (5) window back into the pencil, coupling entry, off-window updates of A, B, Q, Z
        copy_matrix(s, nw, nw, L.dA, ld, A + (size_t)kw * ldA + kw, ldA);
        copy_matrix(s, nw, nw, L.dB, ld, B + (size_t)kw * ldB + kw, ldB);
        if (sub != 0.0)
            hipLaunchKernelGGL(gep_set_entry_kernel, dim3(1), dim3(1), 0, s, A + (size_t)(kw - 1) * ldA + kw, spike[0]);
        apply_transform(kw, nw, L.dQ, L.dZ, ld);
        window_pencil_to_host();                // the caller reads the deflated diagonal blocks from the host copy
        prof_laed[3] += wall() - tl3;
        return res;
    }

    // Q and Z updates of the window steps: issued after the sweep's critical path, on the lazy
    // stream, so that they execute while the host reduces the AED windows that follow
    struct LazyItem { SweepStep step; int ev; };
    std::vector<LazyItem> lazy;
    void flush_lazy()
    {
        if (lazy.empty()) return;
        int const qz = (Q ? 1 : 0) + (Z ? 1 : 0);
        int const fslot = (int)(ws.flush_total % GepWorkspace::FLUSH_RING);
        SN_HIP_CHECK(hipStreamWaitEvent(ws.qs, ws.near_done[lazy.back().ev], 0));
        for (LazyItem const &it : lazy) {
            double *Ubuf = ws.dU + (size_t)it.ev * ws.max_chains * 2 * GWS * GWS;
            hipLaunchKernelGGL(gep_update_kernel<3>, dim3(divceil(n, 128), qz * it.step.ntasks), dim3(256),
                GEP_LDS_BYTES_R, ws.qs, it.step, A, ldA, B, ldB, Q, ldQ, Z, ldZ, n, Ubuf);
            ws.slot_flush[it.ev] = ws.flush_total;
        }
        SN_HIP_CHECK(hipEventRecord(ws.q_done[fslot], ws.qs));
        ws.flush_total++;
        lazy.clear();
    }

    // one multi-shift QZ sweep over the active block [ilo, ihi); same schedule as schur.hip
    void sweep(int ilo, int ihi, int nshifts, double const *sr, double const *si)
    {
        int const size = ihi - ilo;
        int const nbulges = nshifts / 2;
        int const ws_ = std::min(GWS, size);
        int nbc = std::min(GNB, (ws_ - 1) / 6);
        if (nbc < 1) nbc = 1;
        if (size <= GWS) nbc = std::min(nbulges, std::max(1, (size - 1) / 3));
        nbc = std::min(nbc, GNB);
        int const chains = divceil(nbulges, nbc);
        int const adv = ws_ - 1 - 3 * nbc;
        int const gap = (adv > 0) ? divceil(ws_ + adv, adv) : 1;
        SN_HIP_CHECK(hipMemcpyAsync(ws.dShiftR, sr, (size_t)nshifts * 8, hipMemcpyHostToDevice, s));
        SN_HIP_CHECK(hipMemcpyAsync(ws.dShiftI, si, (size_t)nshifts * 8, hipMemcpyHostToDevice, s));
        int const steps_per_chain = (size <= GWS) ? 1 : divceil(size - ws_, adv) + 1;
        int const total_steps = steps_per_chain + (chains - 1) * gap;
        SweepStep step{ilo, ihi, ws_, nbc, adv, gap, nbulges, steps_per_chain, 0, 0, 0};
        hipStream_t const f = ws.far;
        long issued = 0, last_waited_flush = -1;
        int last_t = -2;
        int const qz = (Q ? 1 : 0) + (Z ? 1 : 0);
        for (int t = 0; t < total_steps; t++) {
            int cmin = (t - steps_per_chain + 1 + gap - 1) / gap;
            if (t - steps_per_chain + 1 <= 0) cmin = 0;
            int const cmax = std::min(chains - 1, t / gap);
            if (cmax < cmin) continue;
            step.t = t; step.cmin = cmin; step.ntasks = cmax - cmin + 1;
            int const ntasks = step.ntasks;
            // factor buffers and events live in a ring that runs across sweeps (schur.hip)
            int const ev = (int)(ws.issued_total % ws.ring);
            int const evp = (int)((ws.issued_total + ws.ring - 1) % ws.ring);
            double *Ubuf = ws.dU + (size_t)ev * ws.max_chains * 2 * GWS * GWS;
            if (ws.slot_flush[ev] >= 0) {
                long const fid = ws.slot_flush[ev];
                if (fid != last_waited_flush) {
                    if (ws.flush_total - fid < GepWorkspace::FLUSH_RING)
                        SN_HIP_CHECK(hipStreamWaitEvent(s, ws.q_done[(int)(fid % GepWorkspace::FLUSH_RING)], 0));
                    else {
                        SN_HIP_CHECK(hipEventRecord(ws.lazy_mark, ws.qs));
                        SN_HIP_CHECK(hipStreamWaitEvent(s, ws.lazy_mark, 0));
                    }
                    last_waited_flush = fid;
                }
                ws.slot_flush[ev] = -1;
            }
            // see schur.hip (and tests/test_schur_pipeline.py) for the two wait rules
            bool const serial = tuning().gep_serial;     // debugging aid
            if (issued > 0 && (serial || last_t != t - 1)) SN_HIP_CHECK(hipStreamWaitEvent(s, ws.far_done[evp], 0));
            hipLaunchKernelGGL(gep_chase_kernel, dim3(ntasks), dim3(GEP_CHASE_THREADS), GEP_CHASE_LDS_BYTES, s,
                step, A, ldA, B, ldB, Ubuf, ws.dShiftR, ws.dShiftI);
            st.chase_launches++;
            int max_far = 0, max_lo = 0;
            for (int k = 0; k < ntasks; k++) {
                ChaseTask const tk = make_task(step, k);
                int const rc = n - (tk.lo + tk.n);
                max_far = std::max(max_far, rc - adv);
                max_lo = std::max(max_lo, tk.lo);
                st.gemm_flops += 2.0 * tk.n * tk.n * (2.0 * rc + 2.0 * tk.lo + (double)qz * n);
            }
            if (issued > 0) SN_HIP_CHECK(hipStreamWaitEvent(s, ws.far_done[evp], 0));
            hipLaunchKernelGGL(gep_update_kernel<2>, dim3(1, 2 * ntasks), dim3(256), GEP_LDS_BYTES_L, s,
                step, A, ldA, B, ldB, Q, ldQ, Z, ldZ, n, Ubuf);
            SN_HIP_CHECK(hipEventRecord(ws.near_done[ev], s));
            SN_HIP_CHECK(hipStreamWaitEvent(f, ws.near_done[ev], 0));
            if (max_far > 0)
                hipLaunchKernelGGL(gep_update_kernel<0>, dim3(divceil(max_far, 128), 2 * ntasks), dim3(256),
                    GEP_LDS_BYTES_L, f, step, A, ldA, B, ldB, Q, ldQ, Z, ldZ, n, Ubuf);
            if (max_lo > 0)
                hipLaunchKernelGGL(gep_update_kernel<1>, dim3(divceil(max_lo, 128), 2 * ntasks), dim3(256),
                    GEP_LDS_BYTES_R, f, step, A, ldA, B, ldB, Q, ldQ, Z, ldZ, n, Ubuf);
            SN_HIP_CHECK(hipEventRecord(ws.far_done[ev], f));
            if (qz) lazy.push_back(LazyItem{step, ev});
            issued++;
            ws.issued_total++;
            last_t = t;
            if ((int)lazy.size() >= ws.ring / 2) flush_lazy();
        }
        if (issued > 0)
            SN_HIP_CHECK(hipStreamWaitEvent(s, ws.far_done[(int)((ws.issued_total - 1) % ws.ring)], 0));
        flush_lazy();
        st.sweeps++;
    }
};

} // namespace

int gep_schur_device(hipStream_t caller, int n, double *dA, int ldA, double *dB, int ldB,
    double *dQ, int ldQ, double *dZ, int ldZ, double *real, double *imag, double *beta,
    SchurParams const &prm, SchurStats *stats, int level)
{
    static hipStream_t own = nullptr;
    static hipEvent_t fence = nullptr;
    if (!own) {
        int lo_prio = 0, hi_prio = 0;          // highest priority: see schur.hip
        SN_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio));
        make_stream(&own, true, hi_prio);
        SN_HIP_CHECK(hipEventCreateWithFlags(&fence, hipEventDisableTiming));
    }
    hipStream_t s = own;
    SN_HIP_CHECK(hipEventRecord(fence, caller));
    SN_HIP_CHECK(hipStreamWaitEvent(s, fence, 0));

    // Parameters.  The host QZ kernels cost about 3x the standard ones per window row (O(w^3)),
    // a sweep costs latency and accuracy (error ~ sqrt(number of window multiplications)), and
    // an AED with reordering deflates converged eigenvalues from anywhere in its window.  The
    // measured optimum is a SMALL window with a LOW nibble: at n = 12000 (test driver's random
    // pencil) window 64 / nibble 6 % needs 880 AEDs of 1.8 ms and only 4 sweeps -- 1.9 s and a
    // residual of 74 u, against 7.0 s and 499 u for window 160 / nibble 40 % (137 sweeps).
    // If the AEDs stop deflating (less than 6 % of the window) the sweeps take over as usual.
    int const nw_rule = tuning().gep_window > 0 ? tuning().gep_window : 64;
    // (a window the caller asks for is taken as asked: up to aed_parallel_hard_limit rows by the sequential host
    // kernel, above it by the blocked device path large_aed -- rounds 2-4 silently clamped it to 768)
    int nw_conf = prm.aed_window_size > 0 ? std::min(prm.aed_window_size, n) : std::min(nw_rule, std::max(16, n / 8));
    int const hard_limit = prm.aed_parallel_hard_limit > 0 ? prm.aed_parallel_hard_limit : 300;
    int ns_conf = prm.shift_count > 0 ? prm.shift_count : std::min(100, std::max(2, 2 * nw_conf / 3));
    ns_conf = std::min(ns_conf, 9 * nw_conf / 10);
    ns_conf = std::max(2, ns_conf - ns_conf % 2);
    int const small_limit = prm.small_limit > 0 ? std::min(prm.small_limit, 768)
                                                : std::max(GWS + 32, std::min(200, nw_conf));
    int const nibble = prm.aed_nibble > 0 ? prm.aed_nibble : 6;
    int const iter_limit = prm.iteration_limit > 0 ? prm.iteration_limit : 300;

    GepWorkspace &ws = g_gws[level];
    int const wmax = std::min(n + 8, std::max({nw_conf + nw_conf / 2 + 8, small_limit, 2 * GWS}));
    // Shift multiplicity (see schur.hip): every shift pair of an AED drives `reuse` bulges of the following
    // sweep.  Round 2 left it off for pencils -- at n = 12000 a multiplicity of 4 raised the residual from
    // 460 u to 690 u then.  Since the reflectors are scaled by exact powers of two the extra chain passes
    // cost nothing measurable (general pencil, Hessenberg-triangular reduction + QZ, multiplicity 1 / 4 / 8:
    // n = 8000 QZ 2.18 / 0.99 / 0.74 s, 58 / 16 / 9 sweeps, chain residual 127 / 125 / 132 u; n = 3000
    // 0.40 (2) / 0.29 / 0.24 s, 87 / 85 / 89 u; n = 1500 0.21 / 0.14 s (4), 73 / 68 u), so it is on:
    // 4 below n = 2000, 8 above, from the fifth sweep on, capped so that a sweep carries at most ~450 bulges.
    // SN_GEP_REUSE=k overrides.
    int const reuse_env = tuning().gep_reuse;
    int const reuse_n = n < 2000 ? 4 : 8;
    int const reuse = reuse_env > 0 ? reuse_env : std::max(1, std::min(reuse_n, 450 / std::max(1, ns_conf / 2)));
    ws.ensure(n, wmax, reuse * (ns_conf / 2) + 1);
    GepDriver d{s, n, dA, ldA, dB, ldB, dQ, ldQ, dZ, ldZ, ws, SchurStats{}};
    // the update kernel identifies "no Q" by a null pointer but still needs distinct slots
    hipEvent_t e0, e1;
    SN_HIP_CHECK(hipEventCreate(&e0)); SN_HIP_CHECK(hipEventCreate(&e1));
    SN_HIP_CHECK(hipEventRecord(e0, s));

    // deflation threshold for A's sub-diagonal / AED spike: u*||A||_F (schur/core.c:2390-2436)
    double thres = prm.threshold;
    if (thres == -1.0 || thres == -2.0) {
        double h = 0.0;
        // (fixed summation order: every replica of the reduction gets the same threshold bit for bit)
        sumsq_ordered(s, n, n, dA, ldA, ws.dTmp, ws.dAcc);
        SN_HIP_CHECK(hipMemcpyAsync(&h, ws.dAcc, 8, hipMemcpyDeviceToHost, s));
        SN_HIP_CHECK(hipStreamSynchronize(s));
        thres = DBL_EPSILON * std::sqrt(h);
    } else if (thres == -3.0) thres = 0.0;
    else if (thres < 0.0) return STARNEIG_INVALID_CONFIGURATION;

    // infinity threshold: u*||B||_F by default (schur/core.c:2440-2470)
    double thres_inf = prm.threshold_inf;
    if (thres_inf == -1.0 || thres_inf == -2.0) {
        double h = 0.0;
        sumsq_ordered(s, n, n, dB, ldB, ws.dTmp, ws.dAcc);
        SN_HIP_CHECK(hipMemcpyAsync(&h, ws.dAcc, 8, hipMemcpyDeviceToHost, s));
        SN_HIP_CHECK(hipStreamSynchronize(s));
        thres_inf = DBL_EPSILON * std::sqrt(h);
    } else if (thres_inf <= 0.0) return STARNEIG_INVALID_CONFIGURATION;

    // B-side threshold of the host window kernels (conf->right_threshold, schur/core.c:2438-2449):
    // norm-stable = u*||B||_F, a positive value as given; default and LAPACK (-3) = LAPACK dhgeqz's
    // BTOL of the window at hand (<= 0 tells the kernels to compute it)
    double thres_b = prm.threshold_b;
    if (thres_b == -2.0) {
        double h = 0.0;
        sumsq_ordered(s, n, n, dB, ldB, ws.dTmp, ws.dAcc);
        SN_HIP_CHECK(hipMemcpyAsync(&h, ws.dAcc, 8, hipMemcpyDeviceToHost, s));
        SN_HIP_CHECK(hipStreamSynchronize(s));
        thres_b = DBL_EPSILON * std::sqrt(h);
    } else if (thres_b == -1.0 || thres_b == -3.0) thres_b = 0.0;
    else if (thres_b <= 0.0) return STARNEIG_INVALID_CONFIGURATION;
    d.thres_b = thres_b;

    std::vector<double> sr(8 * wmax), si(8 * wmax), spike(wmax);
    int rc = STARNEIG_SUCCESS;
    int ihi = n, iter = 0, stagnation = 0;
    while (ihi > 0) {
        hipLaunchKernelGGL(gep_scan_subdiag_kernel, dim3(divceil(ihi, 256)), dim3(256),
            0, s, ihi, dA, ldA, thres, ws.dSub, dB, ldB, thres_inf, ws.dSub + n);
        {
            double tw = wall();
            if (ihi > 1)
                SN_HIP_CHECK(hipMemcpyAsync(ws.hSub, ws.dSub, (size_t)(ihi - 1) * 8, hipMemcpyDeviceToHost, s));
            SN_HIP_CHECK(hipMemcpyAsync(ws.hSub + n, ws.dSub + n, (size_t)ihi * 8, hipMemcpyDeviceToHost, s));
            SN_HIP_CHECK(hipStreamSynchronize(s));
            d.st.wait_s += wall() - tw;
        }
        int ilo = ihi - 1;
        while (ilo > 0 && ws.hSub[ilo - 1] != 0.0) ilo--;
        int const size = ihi - ilo;
        // ---- infinite eigenvalues: zeros on B's diagonal inside the active block are chased to its
        // top and deflated (schur/core.c:475-552, cpu_utils.c:360-425, :605-681)
        if (size >= 2) {
            bool any = false;
            for (int i = ilo; i < ihi && !any; i++) any = ws.hSub[n + i] != 0.0;
            if (any) {
                d.push_infinite(ilo, ihi, thres_inf, real, imag, beta);
                stagnation = 0;
                continue;
            }
        }
        if (size == 1 && ws.hSub[n + ilo] != 0.0) {
            // a 1 x 1 block whose B entry is below the infinity threshold of the WHOLE pencil: an infinite
            // eigenvalue (LAPACK dhgeqz zeroes T(ilast, ilast) against its global BTOL before it deflates a
            // 1 x 1 block; the window kernels below only know the norm of their window -- |B(i, i)| itself here)
            SN_HIP_CHECK(hipMemsetAsync(dB + (size_t)ilo * ldB + ilo, 0, sizeof(double), s));
        }
        if (size <= small_limit) {
            int info = d.small_block(ilo, size, real, imag, beta);
            if (info != 0) { rc = STARNEIG_DID_NOT_CONVERGE; break; }
            ihi = ilo; stagnation = 0; continue;
        }
        if (iter >= iter_limit * std::max(1, n / std::max(1, ns_conf))) { rc = STARNEIG_DID_NOT_CONVERGE; break; }

        // ---- aggressive early deflation on the trailing window ---------------------------------
        int nw = std::min(nw_conf, size);
        if (stagnation > 0) nw = std::min(size, std::min(wmax, nw + nw / 20 * stagnation + 2));
        int const kw = ihi - nw;
        double const sub = (kw > ilo) ? ws.hSub[kw - 1] : 0.0;
        int const ldh = GepDriver::host_ld(nw);
        // windows above aed_parallel_hard_limit: the blocked device path (level 0 only: its own windows are small)
        bool const blocked = level == 0 && nw > hard_limit && nw >= 2 * GWS;
        host::AedResult ar;
        double t_aed0 = wall();
        if (blocked) ar = d.large_aed(kw, nw, sub, thres, thres_inf, prm, spike.data(), sr.data(), si.data());
        else {
            d.download_windows(kw, nw);
            t_aed0 = wall();
            ar = host::gep_aed_window(nw, ws.hA, ldh, ws.hB, ldh, ws.hQ, ldh, ws.hZ, ldh, sub, thres,
                spike.data(), sr.data(), si.data(), thres_b);
        }
        d.st.aed_host_s += wall() - t_aed0;
        d.st.aeds++;
        if (ar.failed) { rc = STARNEIG_DID_NOT_CONVERGE; break; }
        if (ar.deflated > 0) {
            if (!blocked) {
                d.upload_windows(kw, nw);
                if (kw > ilo)
                    hipLaunchKernelGGL(gep_set_entry_kernel, dim3(1), dim3(1), 0, s,
                        dA + (size_t)(kw - 1) * ldA + kw, spike[0]);
                d.apply_transform(kw, nw, ws.dQl, ws.dZl, nw);
            }
            SN_HIP_CHECK(hipStreamSynchronize(s));
            if (real) {
                int const off = nw - ar.deflated;
                host::gep_extract_eigenvalues(ar.deflated, ws.hA + (size_t)off * ldh + off, ldh,
                    ws.hB + (size_t)off * ldh + off, ldh, real + ihi - ar.deflated, imag + ihi - ar.deflated,
                    beta + ihi - ar.deflated);
            }
            ihi -= ar.deflated;
            stagnation = 0;
        } else stagnation++;
        if (ihi - ilo <= small_limit) continue;
        if (100 * ar.deflated > nibble * nw) continue;
        int nshifts = std::min(ar.shifts, ns_conf);
        nshifts -= nshifts % 2;
        // exceptional shifts (cf. schur.hip) on the eigenvalue scale a_ii / b_ii
        if (nshifts < 2 || (stagnation > 0 && stagnation % 6 == 0)) {
            int const want = std::max(2, std::min(ns_conf, (ihi - ilo - 2) / 2 * 2));
            int const first = ihi - (want + 2) >= 0 ? ihi - (want + 2) : 0;
            int const cnt = ihi - first;
            std::vector<double> da(cnt), db(cnt);
            SN_HIP_CHECK(hipMemcpy2DAsync(da.data(), 8, dA + (size_t)first * ldA + first, (size_t)(ldA + 1) * 8,
                8, cnt, hipMemcpyDeviceToHost, s));
            SN_HIP_CHECK(hipMemcpy2DAsync(db.data(), 8, dB + (size_t)first * ldB + first, (size_t)(ldB + 1) * 8,
                8, cnt, hipMemcpyDeviceToHost, s));
            SN_HIP_CHECK(hipStreamSynchronize(s));
            nshifts = 0;
            for (int i = ihi - 1; i >= ilo + 2 && nshifts + 2 <= want; i -= 2) {
                double const bi = db[i - first] != 0.0 ? db[i - first] : 1.0;
                double const bm = db[i - 1 - first] != 0.0 ? std::fabs(db[i - 1 - first]) : 1.0;
                double ss = (std::fabs(ws.hSub[i - 1]) + std::fabs(ws.hSub[i - 2])) / bm;
                double const h = da[i - first] / bi;
                double aa = 0.75 * ss + h, bb = ss, cc = -0.4375 * ss, dd = aa, cs, sn;
                host::lanv2(aa, bb, cc, dd, sr[nshifts], si[nshifts], sr[nshifts + 1], si[nshifts + 1], cs, sn);
                if (ss == 0.0) { sr[nshifts] = sr[nshifts + 1] = h + 1e-3 * (1.0 + std::fabs(h)); si[nshifts] = si[nshifts + 1] = 0.0; }
                nshifts += 2;
            }
            if (nshifts < 2) { rc = STARNEIG_DID_NOT_CONVERGE; break; }
        }
        if (stagnation > 60) { rc = STARNEIG_DID_NOT_CONVERGE; break; }

        // (only once sweeps carry the reduction: the reference's ill-conditioned test pencil -- BASELINE config 5,
        // two sweeps between 870 AEDs -- took 5 sweeps and 5 % longer with the multiplicity from the start)
        if (reuse > 1 && iter >= 4 && ihi - ilo > 4 * GWS) {
            for (int r = 1; r < reuse; r++)
                for (int k = 0; k < nshifts; k++) { sr[r * nshifts + k] = sr[k]; si[r * nshifts + k] = si[k]; }
            nshifts *= reuse;
        }
        d.sweep(ilo, ihi, nshifts, sr.data(), si.data());
        iter++;
    }
    SN_HIP_CHECK(hipEventRecord(fence, ws.qs));         // the lazy stream has to drain
    SN_HIP_CHECK(hipStreamWaitEvent(s, fence, 0));
    SN_HIP_CHECK(hipEventRecord(e1, s));
    SN_HIP_CHECK(hipStreamWaitEvent(caller, e1, 0));
    SN_HIP_CHECK(hipEventSynchronize(e1));
    SN_HIP_CHECK(hipEventElapsedTime(&d.st.total_ms, e0, e1));
    SN_HIP_CHECK(hipEventDestroy(e0)); SN_HIP_CHECK(hipEventDestroy(e1));
    if (tuning().schur_profile)
        fprintf(stderr, "[qz level %d] total %.3f s: aed_host %.3f, wait %.3f, push_inf %.3f s for %d infinite eigenvalues (%d windows); n %d sweeps %d aeds %d; "
            "blocked AEDs %d (%d deflation windows): Schur form %.3f, deflation %.3f, restoration %.3f, rest %.3f s\n",
            level, d.st.total_ms * 1e-3, d.st.aed_host_s, d.st.wait_s, d.prof_inf_s, d.st.inf_deflated, d.prof_inf_windows,
            n, d.st.sweeps, d.st.aeds, d.prof_laed_calls, d.prof_laed_windows, d.prof_laed[0], d.prof_laed[1], d.prof_laed[2], d.prof_laed[3]);
    if (stats) *stats = d.st;
    return rc;
}

} // namespace sn
