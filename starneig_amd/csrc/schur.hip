// Multi-shift QR with aggressive early deflation on one MI355X (rows S0-S8 of SURVEY 8a).
//
// Reference: schur/core.c (state machine :2226-2336, AED/bulge policies :1878-1973,
// window placement :668-764), schur/cpu.c + cpu_utils.c (window kernels), common/cpu.c:54-162
// (off-diagonal GEMM updates).  What is kept: the algorithm (small-bulge multi-shift QR
// sweeps, AED with the norm-stable deflation criterion |sub*Z(0,i)| < u*||H||_F, shifts
// ordered/paired as starneig_extract_shifts does, 40 % nibble rule, 2x2 blocks in dlanv2
// standard form, eigenvalues extracted from the diagonal blocks).  What is re-designed for
// the GPU:
//  - H and Q never leave HBM; ONE level of diagonal windows (<= 96 rows, held in LDS together
//    with the accumulated orthogonal factor) instead of the reference's 2*tile-row windows with
//    50x50 sub-windows; all chains of a sweep advance together in one launch per window step
//    (one workgroup per chain); every off-diagonal update is an in-place fp64-MFMA GEMM;
//  - every shift pair drives several bulges (shift multiplicity), the sweeps are few and wide;
//  - updates are split into a timely zone (what later window steps and AED windows read: far
//    stream, one step behind the chase) and lazy zones (Q, the deflated columns, rows above the
//    chain band: low-priority streams, executed while the host is busy);
//  - look-ahead: the head of the next sweep runs above a guard row while the host reduces the
//    chain of AED windows below it on a separate stream;
//  - the sequential small dense problems (AED window, final small blocks) run on the host on
//    copies of the window (schur_host.hip), like the reference's CPU-only window tasks
//    (schur/tasks.c:203-261 have no .cuda_funcs).
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
#include <cstring>
#include <starneig/error.h>

namespace sn {

void sumsq_diff(hipStream_t s, int m, int n, double const *X, int ldx, double const *Y, int ldy,
    double ident, double *acc);
void sumsq_ordered(hipStream_t s, int m, int n, double const *X, int ldx, double *part, double *out);

constexpr int WS_MAX = 96;          // diagonal window (rows) held in LDS
constexpr int NB_MAX = 15;          // bulges per chain: 3*NB_MAX+1 <= WS_MAX/2 + ...
constexpr int LDW = WS_MAX + 1;     // odd leading dimension: conflict-free row AND column walks
constexpr int CHASE_THREADS = 1024; // 16 waves share one window (512 threads: 15 % slower)
constexpr int UPDATE_LDS_BYTES_L = GemmCfg<WS_MAX, 128, 16, true, false>::LDS_BYTES;
constexpr int UPDATE_LDS_BYTES_R = GemmCfg<128, WS_MAX, 16, false, false>::LDS_BYTES;
constexpr int UPDATE_LDS_BYTES_R64 = GemmCfg<64, WS_MAX, 16, false, false>::LDS_BYTES;
constexpr int UPDATE_LDS_BYTES_P = UPDATE_LDS_BYTES_L > UPDATE_LDS_BYTES_R ? UPDATE_LDS_BYTES_L : UPDATE_LDS_BYTES_R;
constexpr int CHASE_LDS_BYTES_WU = (2 * WS_MAX * LDW + 12 * NB_MAX + 16) * 8;   // window W and accumulated factor U in LDS, two reflector buffers (150 KB)

// LAPACK dlaqr1 for a 3x3 block: first column of (H - s1 I)(H - s2 I), scaled
__device__ __forceinline__ void shift_vector(double const *W, double sr1, double si1,
    double sr2, double si2, double *v)
{
    double h11 = W[0], h21 = W[1], h31 = W[2];
    double h12 = W[LDW], h22 = W[LDW + 1], h32 = W[LDW + 2];
    double h13 = W[2 * LDW], h23 = W[2 * LDW + 1], h33 = W[2 * LDW + 2];
    double s = fabs(h11 - sr2) + fabs(si2) + fabs(h21) + fabs(h31);
    if (s == 0.0) { v[0] = v[1] = v[2] = 0.0; return; }
    double h21s = h21 / s, h31s = h31 / s;
    v[0] = (h11 - sr1) * ((h11 - sr2) / s) - si1 * (si2 / s) + h12 * h21s + h13 * h31s;
    v[1] = h21s * (h11 + h22 - sr1 - sr2) + h23 * h31s;
    v[2] = h31s * (h11 + h33 - sr1 - sr2) + h21s * h32;
}

// One workgroup chases one chain of bulges through one diagonal window held in LDS
// (the device counterpart of process_small_window, schur/cpu_utils.c:1168-1810, without
// its QZ branches).  Per column step all lanes apply the step's reflectors (a) from the left,
// (b) from the right to the window and to the accumulated factor U; bulges sit 3 columns apart,
// so the reflectors of one step touch disjoint rows/columns and commute (the LAPACK dlaqr5
// argument).  The reflector of the NEXT step is built inside phase (b): lane i first applies
// bulge i's reflector to the three window rows whose entries in the bulge's first column ARE the
// next reflector's input, builds it from the values it holds in registers and publishes it in
// the other half of a double buffer -- the scalar chain of divisions and a square root runs
// beside the bulk of phase (b) instead of in a phase (and behind a barrier) of its own.
struct ChaseReflector { double v1, v2, tau; int row0, len; };

__device__ __forceinline__ void chase_publish(double *R, int *Ri, int i, double v1, double v2, double tau, int row0, int len)
{
    R[4 * i + 0] = v1; R[4 * i + 1] = v2; R[4 * i + 2] = tau;
    R[4 * i + 3] = (tau != 0.0) ? (double)len : 0.0;   // all four in one 32-byte read
    Ri[2 * i + 0] = row0; Ri[2 * i + 1] = (tau != 0.0) ? len : 0;
}

// reflector of bulge i for the step whose leading column is j (j = -1: introduction from the
// shifts); published in LDS for all lanes and returned in registers for the lane itself
__device__ __forceinline__ ChaseReflector chase_build(double *W, int n, int i, int j, bool have,
    double xi0, double xi1, double xi2, double sr1, double si1, double sr2, double si2, double *R, int *Ri)
{
    int len = 0;
    double beta = 0.0, v1 = 0.0, v2 = 0.0, tau = 0.0;
    if (j >= -1 && j < n - 2) {
        if (j == -1) {
            double x[3];
            shift_vector(W, sr1, si1, sr2, si2, x);
            len = 3;
            small_reflector(3, x, beta, v1, v2, tau);
        } else {
            len = (j == n - 3) ? 2 : 3;
            double *col = W + j * LDW + j + 1;
            // (scalars, not an array handed over by pointer: that would live in scratch memory and put
            // a memory round trip on the serial chain of every column step)
            double x[3];
            x[0] = have ? xi0 : col[0]; x[1] = have ? xi1 : col[1];
            x[2] = (len == 3) ? (have ? xi2 : col[2]) : 0.0;
            small_reflector(len, x, beta, v1, v2, tau);
            col[0] = beta; col[1] = 0.0;
            if (len == 3) col[2] = 0.0;
        }
    }
    chase_publish(R, Ri, i, v1, v2, tau, j + 1, len);
    return ChaseReflector{v1, v2, tau, j + 1, (tau != 0.0) ? len : 0};
}

// (Round 5 built and measured two variants with the accumulated factor in registers instead of LDS -- 75 KB of LDS
// a window, 1024 and 512 threads: bit-identical results, 101 / 122 us a launch alone against 98 us, the Schur leg
// 1.67-1.72 s with all three; profiles/r5_chase_variants.txt, DESIGN.md section 4 -- and round 6 removed them.)
template <int DBG>      // DBG != 0: timing experiments of scratch/chase_bench.py (phases switched off)
__device__ __forceinline__ void schur_chase_body(SweepStep const step, double *__restrict__ H, int ldH,
    double *__restrict__ Uout, double const *__restrict__ sr, double const *__restrict__ si)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *W = lds, *U = lds + WS_MAX * LDW, *Rbase = U + WS_MAX * LDW;   // per bulge {v1,v2,tau,-}, two buffers
    int *Ribase = reinterpret_cast<int *>(Rbase + 8 * NB_MAX);            // per bulge {row0, len}, two buffers
    ChaseTask const t = make_task(step, blockIdx.x);
    int const n = t.n, nb = t.nb, tid = threadIdx.x;
    bool const introduce = t.flags & 1, finalize = t.flags & 2;
    double *Uo = Uout + (size_t)blockIdx.x * WS_MAX * WS_MAX;

    for (int idx = tid; idx < n * n; idx += CHASE_THREADS) {
        int r = idx % n, c = idx / n;
        W[c * LDW + r] = H[(size_t)(t.lo + c) * ldH + t.lo + r];
        U[c * LDW + r] = (r == c) ? 1.0 : 0.0;
    }
    // the (bulge, column) and (bulge, row) pairs a lane owns do not change from step to step
    constexpr int L_ITEMS = (NB_MAX * WS_MAX + CHASE_THREADS - 1) / CHASE_THREADS;
    constexpr int R_ITEMS = (2 * NB_MAX * WS_MAX + CHASE_THREADS - 1) / CHASE_THREADS;
    int const rspan = 2 * n;                                             // items per bulge of the right phase: window rows, rows of U
    int li[L_ITEMS], lc[L_ITEMS], ri[R_ITEMS], rr_[R_ITEMS];
    #pragma unroll
    for (int k = 0; k < L_ITEMS; k++) {
        int const item = tid + k * CHASE_THREADS;
        li[k] = item < nb * n ? item / n : -1; lc[k] = item - (item / n) * n;
    }
    #pragma unroll
    for (int k = 0; k < R_ITEMS; k++) {
        int const item = tid + k * CHASE_THREADS;
        ri[k] = item < nb * rspan ? item / rspan : -1; rr_[k] = item - (item / rspan) * rspan;
    }
    __syncthreads();

    int const left = introduce ? 2 - 3 * nb : 0;
    int const right = finalize ? n - 2 : t.right;
    ChaseReflector mine{0.0, 0.0, 0.0, 0, 0};
    double sr1 = 0.0, si1 = 0.0, sr2 = 0.0, si2 = 0.0;         // the lane's shift pair (used at introduction)
    if (tid < nb && introduce) {
        sr1 = sr[t.shift_off + 2 * tid]; si1 = si[t.shift_off + 2 * tid];
        sr2 = sr[t.shift_off + 2 * tid + 1]; si2 = si[t.shift_off + 2 * tid + 1];
    }
    if (tid < nb && left < right) mine = chase_build(W, n, tid, left + 3 * tid, false, 0.0, 0.0, 0.0, sr1, si1, sr2, si2, Rbase, Ribase);
    __syncthreads();
    for (int begin = left; begin < right; begin++) {
        int const cur = (begin - left) & 1;
        double const *R = Rbase + cur * 4 * NB_MAX;
        // The position of bulge i is arithmetic (row0 = begin + 3 i + 1), so an item's window entries
        // and its reflector are fetched in ONE round of LDS reads (clamped addresses, predicated
        // stores) instead of a chain of dependent reads -- the phases are LDS-latency-bound.
        // (a) left: rows row0..row0+len-1, columns max(row0,0)..n-1
        #pragma unroll
        for (int k = 0; k < L_ITEMS; k++) {
            int const i = li[k], c = lc[k];
            if (i < 0 || (DBG & 1)) continue;
            int const row0 = begin + 3 * i + 1;
            int const rs = max(0, min(max(row0, 0), n - 3));
            double *p = W + c * LDW + rs;
            double const x0 = p[0], x1 = p[1], x2r = p[2];
            d4 const rf = *reinterpret_cast<d4 const *>(R + 4 * i);      // v1, v2, tau, len
            int const len = (int)rf.w;
            if (len == 0 || c < row0) continue;                          // (len != 0 implies 0 <= row0 <= n-2)
            double *q = W + c * LDW + row0;
            double const y2 = (len == 3) ? x2r : 0.0;
            double const y0 = (rs == row0) ? x0 : q[0], y1 = (rs == row0) ? x1 : q[1];   // len == 2 at the window end
            double const sum = rf.z * (y0 + rf.x * y1 + rf.y * y2);
            q[0] = y0 - sum; q[1] = y1 - sum * rf.x;
            if (len == 3) q[2] = y2 - sum * rf.y;
        }
        __syncthreads();
        // (b) right: columns row0..row0+len-1; window rows 0..min(n-1,row0+3), all rows of U.
        // Lane i < nb owns the window rows row0+1..row0+3 of bulge i and builds the next reflector.
        if (tid < nb && !(DBG & 4)) {
            int const i = tid;
            int const len = mine.len, row0 = mine.row0;
            double xa = 0.0, xb = 0.0, xc = 0.0;
            bool have = false;
            if (len != 0) {
                double const v1 = mine.v1, v2 = mine.v2, tau = mine.tau;
                // all nine entries in one round of reads (rows clamped to the window)
                int const r0 = min(row0 + 1, n - 1), r1 = min(row0 + 2, n - 1), r2 = min(row0 + 3, n - 1);
                double *base = W + row0 * LDW;
                int const c2 = (len == 3) ? 2 * LDW : LDW;
                double a0 = base[r0], a1 = base[LDW + r0], a2 = base[c2 + r0];
                double b0 = base[r1], b1 = base[LDW + r1], b2 = base[c2 + r1];
                double c0 = base[r2], c1 = base[LDW + r2], c2v = base[c2 + r2];
                if (len != 3) { a2 = 0.0; b2 = 0.0; c2v = 0.0; }
                double const sa = tau * (a0 + v1 * a1 + v2 * a2);
                double const sb = tau * (b0 + v1 * b1 + v2 * b2);
                double const sc = tau * (c0 + v1 * c1 + v2 * c2v);
                xa = a0 - sa; xb = b0 - sb; xc = c0 - sc;
                if (row0 + 1 <= n - 1) { base[r0] = xa; base[LDW + r0] = a1 - sa * v1; if (len == 3) base[2 * LDW + r0] = a2 - sa * v2; }
                if (row0 + 2 <= n - 1) { base[r1] = xb; base[LDW + r1] = b1 - sb * v1; if (len == 3) base[2 * LDW + r1] = b2 - sb * v2; }
                if (row0 + 3 <= n - 1) { base[r2] = xc; base[LDW + r2] = c1 - sc * v1; if (len == 3) base[2 * LDW + r2] = c2v - sc * v2; }
                have = true;
            }
            if (begin + 1 < right)
                mine = chase_build(W, n, i, begin + 1 + 3 * i, have, xa, xb, xc, sr1, si1, sr2, si2,
                    Rbase + (cur ^ 1) * 4 * NB_MAX, Ribase + (cur ^ 1) * 2 * NB_MAX);
        }
        #pragma unroll
        for (int k = 0; k < R_ITEMS; k++) {
            int const i = ri[k], rr = rr_[k];
            if (i < 0 || (DBG & 2)) continue;
            int const row0 = begin + 3 * i + 1;
            int const cs = max(0, min(max(row0, 0), n - 3));            // clamped column for the reads
            bool const inW = rr < n;
            int const r = inW ? rr : rr - n;
            double *M = inW ? W : U;
            double *p = M + cs * LDW + r;
            double const x0 = p[0], x1 = p[LDW], x2r = p[2 * LDW];
            d4 const rf = *reinterpret_cast<d4 const *>(R + 4 * i);
            int const len = (int)rf.w;
            if (len == 0 || (inW && r > row0)) continue;                 // rows row0+1.. belong to lane i
            double *q = M + row0 * LDW + r;
            double const y2 = (len == 3) ? x2r : 0.0;
            double const y0 = (cs == row0) ? x0 : q[0], y1 = (cs == row0) ? x1 : q[LDW];
            double const sum = rf.z * (y0 + rf.x * y1 + rf.y * y2);
            q[0] = y0 - sum; q[LDW] = y1 - sum * rf.x;
            if (len == 3) q[2 * LDW] = y2 - sum * rf.y;
        }
        __syncthreads();
    }

    for (int idx = tid; idx < n * n; idx += CHASE_THREADS) {
        int r = idx % n, c = idx / n;
        H[(size_t)(t.lo + c) * ldH + t.lo + r] = W[c * LDW + r];
        Uo[c * WS_MAX + r] = U[c * LDW + r];
    }
}

// the window AND the accumulated factor in LDS (150 KB): one workgroup per chain
__global__ __launch_bounds__(CHASE_THREADS)
void schur_chase_ulds_kernel(SweepStep const step, double *__restrict__ H, int ldH,
    double *__restrict__ Uout, double const *__restrict__ sr, double const *__restrict__ si)
{
    __builtin_amdgcn_s_setprio(3);      // the latency-bound chain outranks the update kernels in instruction issue
    schur_chase_body<0>(step, H, ldH, Uout, sr, si);
}
#ifdef SN_TEST_HOOKS
template <int DBG>
__global__ __launch_bounds__(CHASE_THREADS)
void schur_chase_dbg_kernel(SweepStep const step, double *__restrict__ H, int ldH,
    double *__restrict__ Uout, double const *__restrict__ sr, double const *__restrict__ si)
{
    schur_chase_body<DBG>(step, H, ldH, Uout, sr, si);
}
#endif

// Off-diagonal updates of all chains of one step (row S3).
//   MODE 2 ("near"): H(win, next `adv` columns right of win) <- U^T .  -- the only part of the
//                    updates the chain's NEXT window needs; stays on the critical stream.
//   MODE 0 ("far") : the remaining columns right of the window.
//   MODE 1         : H(above win, win) <- . U
//   MODE 3         : Q(:, win) <- . U
// MODE 0/1 run on a second stream concurrently with the next chase launch (chains are
// spaced ws+adv rows apart so that no other chain's next window touches them).
// Timely and lazy zones: the iteration only ever reads H inside the band of rows between the
// rearmost chain and the bottom of the active block, and in columns left of ihi.  Updates of
//   - Q,
//   - H(:, ihi:n)          (the columns of the already deflated part, MODE 0), and
//   - H(0:T0, :)           (rows above the rearmost chain and above the guard row R1, MODE 1)
// are "lazy": they run in issue order on a third, low-priority stream and fill the time the GPU
// would otherwise idle while the host reduces AED windows.  [r0, r1) selects the column
// (MODE 0) or row (MODE 1) range of a launch.
// One workgroup owns all w <= WS_MAX = 96 rows (columns) of its tile (96-wide MFMA tiles: no padding) and reads its whole operand
// panel before the epilogue writes, so the update is done in place.
template <int MODE, int RBM = 128>
__device__ __forceinline__
void schur_update_body(SweepStep const &step, double *__restrict__ H, int ldH,
    double *__restrict__ Q, int ldQ, int n, double const *__restrict__ U, int r0, int r1, int bx, int by)
{
    int const k = by % step.ntasks;
    ChaseTask const t = make_task(step, k);
    double const *Uk = U + (size_t)k * WS_MAX * WS_MAX;
    int const w = t.n, lo = t.lo;
    if (MODE == 0 || MODE == 2) {
        // MODE 2: columns [lo+w, lo+w+adv) left of ihi (the deflated columns are lazy);
        // MODE 0: the columns right of the near strip (timely launch, r1 <= ihi) or right of the
        //         window (lazy launch, r0 >= ihi) that fall into [r0, r1)
        int c0 = lo + w, cend = n;
        if (MODE == 2) cend = min(step.ihi, c0 + step.adv);
        else {
            if (r0 < step.ihi) c0 += step.adv;
            c0 = max(c0, r0); cend = min(cend, r1);
        }
        int const ncols = cend - c0;
        if (bx * 128 >= ncols) return;
        double *X = H + (size_t)c0 * ldH + lo;
        gemm_tile<WS_MAX, 128, 16, true, false>(w, ncols, w, 1.0, Uk, WS_MAX, X, ldH, 0.0, X, ldH, 0, bx);
    } else if (MODE == 1) {
        // the rows of [0, lo) that fall into [r0, r1)
        int const rbeg = r0, rows = min(lo, r1) - rbeg;
        if (bx * 128 >= rows) return;
        double *X = H + (size_t)lo * ldH + rbeg;
        gemm_tile<128, WS_MAX, 16, false, false>(rows, w, w, 1.0, X, ldH, Uk, WS_MAX, 0.0, X, ldH, bx, 0);
    } else {
        if (bx * RBM >= n) return;
        double *X = Q + (size_t)lo * ldQ;
        gemm_tile<RBM, WS_MAX, 16, false, false>(n, w, w, 1.0, X, ldQ, Uk, WS_MAX, 0.0, X, ldQ, bx, 0);
    }
}

template <int MODE, int RBM = 128>
__global__ __launch_bounds__(256, 2)
void schur_update_kernel(SweepStep const step, double *__restrict__ H, int ldH,
    double *__restrict__ Q, int ldQ, int n, double const *__restrict__ U, int r0, int r1)
{
    if (MODE != 3) __builtin_amdgcn_s_setprio(2);      // timely updates (the lazy ones keep priority 0)
    schur_update_body<MODE, RBM>(step, H, ldH, Q, ldQ, n, U, r0, r1, blockIdx.x, blockIdx.y);
}

// The LAZY far-left (blockIdx.z = 0, columns [c0, c1) right of ihi) and right (blockIdx.z = 1,
// rows [r0, r1) above the band) updates of one step in one launch -- they touch disjoint
// columns.  (The host issues ~10^4 steps per reduction and its launch rate, not the GPU,
// bounds the reduction when every part is a launch of its own.)
__global__ __launch_bounds__(256, 2)
void schur_update_pair_kernel(SweepStep const step, double *__restrict__ H, int ldH, int n,
    double const *__restrict__ U, int c0, int c1, int r0, int r1)
{
    if (blockIdx.z == 0) schur_update_body<0>(step, H, ldH, nullptr, 0, n, U, c0, c1, blockIdx.x, blockIdx.y);
    else schur_update_body<1>(step, H, ldH, nullptr, 0, n, U, r0, r1, blockIdx.x, blockIdx.y);
}

// The same launch when `world` GPUs reduce replicas of H (the sharded Schur leg).  The columns right of
// ihi belong to the deflated part: no later window reads or mixes them, they only ever receive further
// left updates, and each of their entries depends on its own column alone.  So a 128-column tile of
// them (global tile index T = column / 128, T >= ceil(ihi / 128)) is kept up to date by ONE rank,
// T % world, and the others skip it -- no exchange during the reduction; the caller assembles H from
// the owners' tiles at the end (tile T from rank T % world; ihi only shrinks, so a tile that became
// owner-only stays so).  Columns between the window and the first whole deflated tile stay replicated.
// blockIdx.x < rep_tiles: the replicated strip; beyond: the rank's own tiles, `world` apart.
__global__ __launch_bounds__(256, 2)
void schur_update_pair_sharded_kernel(SweepStep const step, double *__restrict__ H, int ldH, int n,
    double const *__restrict__ U, int c0, int c1, int r0, int r1, int rep_tiles, int rank, int world)
{
    if (blockIdx.z != 0) {
        schur_update_body<1>(step, H, ldH, nullptr, 0, n, U, r0, r1, blockIdx.x, blockIdx.y);
        return;
    }
    int const first_own = (step.ihi + 127) / 128;                  // first whole tile of deflated columns
    int const bx = blockIdx.x;
    if (bx < rep_tiles) {
        schur_update_body<0>(step, H, ldH, nullptr, 0, n, U, c0, min(c1, first_own * 128), bx, blockIdx.y);
        return;
    }
    int T = first_own + (rank - first_own % world + world) % world + (bx - rep_tiles) * world;
    int const t0 = max(c0, T * 128), t1 = min(c1, T * 128 + 128);
    if (t0 >= t1) return;
    // the window's own columns and its near strip never reach here: T * 128 >= ihi >= lo + w
    schur_update_body<0>(step, H, ldH, nullptr, 0, n, U, t0, t1, 0, blockIdx.y);
}

// sub[i] = H(i+1,i) for i in [lo,hi-1); entries below the threshold are set to exactly zero
// in H as well (the small-sub-diagonal deflation of schur/core.c:1834-1856 / vigilant
// deflation with the norm-stable criterion).  thres <= 0 selects the LAPACK criterion.
__global__ void schur_scan_subdiag_kernel(int lo, int hi, double *__restrict__ H, int ldH,
    double thres, double *__restrict__ sub, int n)
{
    int i = lo + blockIdx.x * 256 + threadIdx.x;
    if (i >= hi - 1) return;
    double *p = H + (size_t)i * ldH + i + 1;
    double v = *p;
    if (v != 0.0) {
        bool small;
        if (thres > 0.0) small = fabs(v) < thres;
        else {
            double const ulp = DBL_EPSILON, smlnum = DBL_MIN * ((double)n / ulp);
            double a = H[(size_t)i * ldH + i], d = H[(size_t)(i + 1) * ldH + i + 1];
            double tst = fabs(a) + fabs(d);
            small = fabs(v) <= fmax(smlnum, ulp * tst);
            if (small) {
                double b = H[(size_t)(i + 1) * ldH + i];
                double ab = fmax(fabs(v), fabs(b)), ba = fmin(fabs(v), fabs(b));
                double aa = fmax(fabs(d), fabs(a - d)), bb = fmin(fabs(d), fabs(a - d));
                double s = aa + ab;
                small = ba * (ab / s) <= fmax(smlnum, ulp * (bb * (aa / s)));
            }
        }
        if (small) { v = 0.0; *p = 0.0; }
    }
    sub[i] = v;
}

__global__ void schur_set_entry_kernel(double *p, double v) { *p = v; }

// ---- workspace --------------------------------------------------------------------------
struct SchurWorkspace {
    int n = 0, nwmax = 0, max_chains = 0;
    double *dU = nullptr;           // max_chains x WS_MAX x WS_MAX
    double *dShiftR = nullptr, *dShiftI = nullptr;
    double *dSub = nullptr;         // n
    double *dWin = nullptr, *dZ = nullptr, *dTmp = nullptr;   // nwmax^2, nwmax^2, n*nwmax
    double *dAcc = nullptr;
    double *hWin = nullptr, *hZ = nullptr, *hSub = nullptr, *hShift = nullptr;   // pinned
    long shift_uploads = 0;
    bool attr_set = false;
    hipStream_t far = nullptr, qs = nullptr, hs = nullptr;    // timely far H updates; lazy Q; lazy H
    hipStream_t aed = nullptr;      // window traffic and timely AED updates while a sweep head is in flight
    hipEvent_t aed_mark = nullptr;
    static constexpr int EV_RING = 2048;        // largest ring of per-step U buffers (the lazy streams lag a sweep)
    int ring = EV_RING;                         // ring in use: ~4 sweeps' worth of window steps
    hipEvent_t near_done[EV_RING] = {}, far_done[EV_RING] = {};
    static constexpr int FLUSH_RING = 256;       // lazy launches are issued in batches; one event pair per batch
    hipEvent_t q_done[FLUSH_RING] = {}, h_done[FLUSH_RING] = {};
    long flush_total = 0;
    std::vector<long> slot_flush = std::vector<long>(EV_RING, -1);     // batch that consumes the U slot
    static constexpr int Z_RING = 64;           // AED / small-block factors waiting for the lazy stream
    double *dZq = nullptr, *dTmpQ = nullptr;
    hipEvent_t z_ready[Z_RING] = {}, z_done[Z_RING] = {}, zh_done[Z_RING] = {};
    double *dTmpH = nullptr;
    long issued_total = 0, z_total = 0;
    int guard_row = 0;          // R1: rows >= guard_row are always updated timely
    hipEvent_t lazy_mark = nullptr;

    void release() {
        void **dptrs[] = {(void **)&dU, (void **)&dShiftR, (void **)&dShiftI,
            (void **)&dSub, (void **)&dWin, (void **)&dZ, (void **)&dTmp, (void **)&dAcc, (void **)&dZq, (void **)&dTmpQ, (void **)&dTmpH};
        for (auto p : dptrs) if (*p) { SN_HIP_CHECK(hipFree(*p)); *p = nullptr; }
        void **hptrs[] = {(void **)&hWin, (void **)&hZ, (void **)&hSub, (void **)&hShift};
        for (auto p : hptrs) if (*p) { SN_HIP_CHECK(hipHostFree(*p)); *p = nullptr; }
        n = nwmax = max_chains = 0;
    }
    // streams and events go when the owning thread lets the workspace go (node finalize, the end of a team
    // thread): a stream holds a reference on a hardware queue -- or, created with a CU mask, the queue itself
    void destroy_streams() {
        if (!attr_set) return;
        auto kill = [](hipEvent_t &e) { if (e) { SN_HIP_CHECK(hipEventDestroy(e)); e = nullptr; } };
        for (hipStream_t *st : {&far, &qs, &hs, &aed}) if (*st) { SN_HIP_CHECK(hipStreamDestroy(*st)); *st = nullptr; }
        for (int k = 0; k < EV_RING; k++) { kill(near_done[k]); kill(far_done[k]); }
        for (int k = 0; k < FLUSH_RING; k++) { kill(q_done[k]); kill(h_done[k]); }
        for (int k = 0; k < Z_RING; k++) { kill(z_ready[k]); kill(z_done[k]); kill(zh_done[k]); }
        kill(lazy_mark); kill(aed_mark);
        // the rings restart with the fresh events
        flush_total = 0; issued_total = 0; z_total = 0; shift_uploads = 0;
        std::fill(slot_flush.begin(), slot_flush.end(), -1L);
        attr_set = false;
    }
    void ensure(int n_, int nw_, int chains_) {
        if (n_ <= n && nw_ <= nwmax && chains_ <= max_chains) return;
        release();
        n = n_; nwmax = nw_; max_chains = chains_;
        ring = std::min(EV_RING, std::max(64, 4 * (n / 40 + 128)));
        std::fill(slot_flush.begin(), slot_flush.end(), -1L);
        SN_HIP_CHECK(hipMalloc((void **)&dU, (size_t)ring * max_chains * WS_MAX * WS_MAX * 8));
        SN_HIP_CHECK(hipMalloc((void **)&dZq, (size_t)Z_RING * nwmax * nwmax * 8));
        SN_HIP_CHECK(hipMalloc((void **)&dTmpQ, (size_t)n * nwmax * 8));
        SN_HIP_CHECK(hipMalloc((void **)&dTmpH, (size_t)n * nwmax * 8));
        SN_HIP_CHECK(hipMalloc((void **)&dShiftR, (size_t)8 * nwmax * 8));
        SN_HIP_CHECK(hipMalloc((void **)&dShiftI, (size_t)8 * nwmax * 8));
        SN_HIP_CHECK(hipMalloc((void **)&dSub, (size_t)n * 8));
        SN_HIP_CHECK(hipMalloc((void **)&dWin, (size_t)nwmax * nwmax * 8));
        SN_HIP_CHECK(hipMalloc((void **)&dZ, (size_t)nwmax * nwmax * 8));
        SN_HIP_CHECK(hipMalloc((void **)&dTmp, (size_t)n * nwmax * 8));
        SN_HIP_CHECK(hipMalloc((void **)&dAcc, 4 * 8));
        SN_HIP_CHECK(hipHostMalloc((void **)&hWin, (size_t)(nwmax + 24) * nwmax * 8, hipHostMallocDefault));
        SN_HIP_CHECK(hipHostMalloc((void **)&hZ, (size_t)(nwmax + 24) * nwmax * 8, hipHostMallocDefault));
        SN_HIP_CHECK(hipHostMalloc((void **)&hSub, (size_t)n * 8, hipHostMallocDefault));
        SN_HIP_CHECK(hipHostMalloc((void **)&hShift, (size_t)4 * 8 * nwmax * 8, hipHostMallocDefault));
        if (!attr_set) {
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)schur_chase_ulds_kernel,
                hipFuncAttributeMaxDynamicSharedMemorySize, CHASE_LDS_BYTES_WU));
            int lo_prio = 0, hi_prio = 0;
            SN_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio));
            make_stream(&far, true, hi_prio);
            int const keep_free = tuning().schur_cumask;
            if (keep_free > 0) {
                // experiment: the lazy update streams may not use the first `keep_free` CUs
                hipDeviceProp_t prop; int dev = 0;
                SN_HIP_CHECK(hipGetDevice(&dev)); SN_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
                int const ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
                std::vector<uint32_t> mask(words, 0u);
                for (int i = keep_free; i < ncu; i++) mask[i / 32] |= 1u << (i % 32);
                SN_HIP_CHECK(hipExtStreamCreateWithCUMask(&qs, words, mask.data()));
                SN_HIP_CHECK(hipExtStreamCreateWithCUMask(&hs, words, mask.data()));
            } else {
                make_stream(&qs, false, lo_prio, tuning().stream_lazy_free);
                // the lazy H stream -- the one the critical stream waits for at the start of every sweep --
                // one level above the lazy Q stream, which nobody waits for before the end (measured at
                // n = 20000: 2.60 s against 2.70 s with both at the lowest priority)
                int const hs_prio = tuning().schur_hs_prio ? std::max(hi_prio, lo_prio - 1) : lo_prio;
                make_stream(&hs, false, hs_prio, tuning().stream_lazy_free);
            }
            for (int k = 0; k < EV_RING; k++) {
                SN_HIP_CHECK(hipEventCreateWithFlags(&near_done[k], hipEventDisableTiming));
                SN_HIP_CHECK(hipEventCreateWithFlags(&far_done[k], hipEventDisableTiming));
            }
            for (int k = 0; k < FLUSH_RING; k++) {
                SN_HIP_CHECK(hipEventCreateWithFlags(&q_done[k], hipEventDisableTiming));
                SN_HIP_CHECK(hipEventCreateWithFlags(&h_done[k], hipEventDisableTiming));
            }
            SN_HIP_CHECK(hipEventCreateWithFlags(&lazy_mark, hipEventDisableTiming));
            // (high priority: a hardware-queue pool of its own, see schur_device)
            make_stream(&aed, true, hi_prio);
            SN_HIP_CHECK(hipEventCreateWithFlags(&aed_mark, hipEventDisableTiming));
            for (int k = 0; k < Z_RING; k++) {
                SN_HIP_CHECK(hipEventCreateWithFlags(&z_ready[k], hipEventDisableTiming));
                SN_HIP_CHECK(hipEventCreateWithFlags(&z_done[k], hipEventDisableTiming));
                SN_HIP_CHECK(hipEventCreateWithFlags(&zh_done[k], hipEventDisableTiming));
            }
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)schur_update_kernel<3, 64>,
                hipFuncAttributeMaxDynamicSharedMemorySize, UPDATE_LDS_BYTES_R64));
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)schur_update_kernel<3>,
                hipFuncAttributeMaxDynamicSharedMemorySize, UPDATE_LDS_BYTES_R));
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)schur_update_pair_kernel,
                hipFuncAttributeMaxDynamicSharedMemorySize, UPDATE_LDS_BYTES_P));
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)schur_update_pair_sharded_kernel,
                hipFuncAttributeMaxDynamicSharedMemorySize, UPDATE_LDS_BYTES_P));
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)schur_update_kernel<2>,
                hipFuncAttributeMaxDynamicSharedMemorySize, UPDATE_LDS_BYTES_L));
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)schur_update_kernel<0>,
                hipFuncAttributeMaxDynamicSharedMemorySize, UPDATE_LDS_BYTES_L));
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)schur_update_kernel<1>,
                hipFuncAttributeMaxDynamicSharedMemorySize, UPDATE_LDS_BYTES_R));
            attr_set = true;
        }
    }
};
// level 0: the caller's matrix; level 1: the private AED window of a blocked AED (large_aed)
static thread_local SchurWorkspace g_sws[2];      // per host thread (= per device of the in-process multi-GPU path)

// private matrices of the blocked AED (row S5): the window T, its Schur vectors Z, the padded
// matrix of the re-Hessenberg step and its transformation
struct LargeAedBuffers {
    int cap = 0, ld = 0;
    double *dT = nullptr, *dZ = nullptr, *dP = nullptr, *dQp = nullptr, *dZl = nullptr, *dTmp = nullptr;
    void release() {
        double **p[] = {&dT, &dZ, &dP, &dQp, &dZl, &dTmp};
        for (auto q : p) if (*q) { SN_HIP_CHECK(hipFree(*q)); *q = nullptr; }
        cap = 0;
    }
    void ensure(int nw) {
        if (nw <= cap) return;
        release();
        cap = nw; ld = (int)roundup((size_t)nw + 1, 16);
        size_t const bytes = (size_t)ld * ld * 8;
        SN_HIP_CHECK(hipMalloc((void **)&dT, bytes)); SN_HIP_CHECK(hipMalloc((void **)&dZ, bytes));
        SN_HIP_CHECK(hipMalloc((void **)&dP, bytes)); SN_HIP_CHECK(hipMalloc((void **)&dQp, bytes));
        SN_HIP_CHECK(hipMalloc((void **)&dTmp, bytes));
        SN_HIP_CHECK(hipMalloc((void **)&dZl, (size_t)128 * 128 * 8));
    }
};
static thread_local LargeAedBuffers g_large;
// the reduction's own stream per recursion level (schur_device) and the event that orders it behind the caller's
static thread_local hipStream_t own_[2] = {nullptr, nullptr};
static thread_local hipEvent_t fence_[2] = {nullptr, nullptr};
void schur_release_workspace()
{
    for (int l = 0; l < 2; l++) {
        g_sws[l].release(); g_sws[l].destroy_streams();
        if (own_[l]) { SN_HIP_CHECK(hipStreamDestroy(own_[l])); own_[l] = nullptr; }
        if (fence_[l]) { SN_HIP_CHECK(hipEventDestroy(fence_[l])); fence_[l] = nullptr; }
    }
    g_large.release();
}

// LAPACK iparmq-style minimum, then the reference's rules (schur/process_args.c:116-162)
static int lapack_min_shifts(int n)
{
    if (n < 30) return 2;
    if (n < 60) return 4;
    if (n < 150) return 10;
    if (n < 590) { double x = (n - 150) / (590 - 150); return (int)((1 - x) * 10 + x * 64); }
    if (n < 3000) return 64;
    if (n < 6000) return 128;
    return 256;
}

namespace {

static inline double wall()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Driver {
    hipStream_t s;
    int n; double *H; int ldH; double *Q; int ldQ;
    SchurWorkspace &ws;
    SchurStats st;
    // stream of the window downloads/uploads and of the timely AED / small-block updates: the
    // critical stream, or the AED stream while the head of the next sweep is in flight on s
    hipStream_t ts = nullptr;
    // rows of Q this process updates (Q points at the first of them): all n, or the row block of
    // a rank when several GPUs reduce replicas of H and share the accumulation of Q
    int nq = 0;
    // ... and the rank / number of those GPUs: the deflated columns of H are kept up to date tile by tile
    // by their owners only (schur_update_pair_sharded_kernel)
    int shard_rank = 0, shard_world = 1;

    // the critical stream waits until the lazy H updates issued so far are done
    void wait_lazy_h()
    {
        SN_HIP_CHECK(hipEventRecord(ws.lazy_mark, ws.hs));
        SN_HIP_CHECK(hipStreamWaitEvent(s, ws.lazy_mark, 0));
    }

    // Moves the guard row R1 (rows >= R1 are updated timely).  Rows that change from the lazy
    // to the timely zone must first see the lazy updates issued so far.
    void set_guard_row(int r1)
    {
        bool const norows = tuning().schur_nolazyrows;   // debugging aid
        r1 = norows ? 0 : std::max(0, r1);
        if (r1 < ws.guard_row) { wait_lazy_h(); prof_guard_moves++; }
        ws.guard_row = r1;
    }

    // X(rows r0..r1 of the column block at `lo`) <- . Z   on stream st (scratch tmp for w > 128)
    void right_update(hipStream_t st, double *M, int ldM, int r0, int r1, int lo, int w,
        double const *dZ, int ldz, double *tmp)
    {
        int const rows = r1 - r0;
        if (rows <= 0) return;
        double *X = M + (size_t)lo * ldM + r0;
        if (w <= 128) dgemm_right_inplace(st, rows, w, dZ, ldz, X, ldM);
        else {
            dgemm(st, 'N', 'N', rows, w, w, 1.0, X, ldM, dZ, ldz, 0.0, tmp, rows);
            copy_matrix(st, rows, w, tmp, rows, X, ldM);
        }
    }

    // H(lo:lo+w, lo+w:n) <- Z^T .,  H(0:lo, lo:lo+w) <- . Z,  Q(:, lo:lo+w) <- . Z
    // (insert_updates of schur/core.c:129-460, bodies common/cpu.c:54-162).  The window is the
    // trailing part of the active block, so the columns right of it belong to the deflated part:
    // that update, the rows above the guard row and Q go to the lazy stream.
    void apply_transform(int lo, int w, double const *dZ, int ldz)
    {
        int const right_cols = n - (lo + w);
        int const split = std::min(lo, ws.guard_row);
        // timely: H(split:lo, window columns)
        right_update(ts, H, ldH, split, lo, lo, w, dZ, ldz, ws.dTmp);
        // lazy, from a private copy of Z (the caller's buffer is reused by the next AED long
        // before the lazy stream gets here)
        int const slot = (int)(ws.z_total % SchurWorkspace::Z_RING);
        double *Zc = ws.dZq + (size_t)slot * ws.nwmax * ws.nwmax;
        if (ws.z_total >= SchurWorkspace::Z_RING) {
            SN_HIP_CHECK(hipStreamWaitEvent(ts, ws.z_done[slot], 0));
            SN_HIP_CHECK(hipStreamWaitEvent(ts, ws.zh_done[slot], 0));
        }
        copy_matrix(ts, w, w, dZ, ldz, Zc, w);
        SN_HIP_CHECK(hipEventRecord(ws.z_ready[slot], ts));
        SN_HIP_CHECK(hipStreamWaitEvent(ws.hs, ws.z_ready[slot], 0));
        if (right_cols > 0) {
            double *X = H + (size_t)(lo + w) * ldH + lo;
            if (w <= 128) dgemm_left_inplace(ws.hs, w, right_cols, Zc, w, X, ldH);
            else {
                dgemm(ws.hs, 'T', 'N', w, right_cols, w, 1.0, Zc, w, X, ldH, 0.0, ws.dTmpH, w);
                copy_matrix(ws.hs, w, right_cols, ws.dTmpH, w, X, ldH);
            }
        }
        right_update(ws.hs, H, ldH, 0, split, lo, w, Zc, w, ws.dTmpH);
        SN_HIP_CHECK(hipEventRecord(ws.zh_done[slot], ws.hs));
        SN_HIP_CHECK(hipStreamWaitEvent(ws.qs, ws.z_ready[slot], 0));
        if (Q) right_update(ws.qs, Q, ldQ, 0, nq, lo, w, Zc, w, ws.dTmpQ);
        SN_HIP_CHECK(hipEventRecord(ws.z_done[slot], ws.qs));
        ws.z_total++;
        st.gemm_flops += 2.0 * w * w * ((double)right_cols + lo + (Q ? nq : 0));
    }

    void download_window(int lo, int w, double *h, int ldh)
    {
        double t0 = wall();
        SN_HIP_CHECK(hipMemcpy2DAsync(h, (size_t)ldh * 8, H + (size_t)lo * ldH + lo, (size_t)ldH * 8,
            (size_t)w * 8, w, hipMemcpyDeviceToHost, ts));
        SN_HIP_CHECK(hipStreamSynchronize(ts));
        st.wait_s += wall() - t0; prof_dl_wait += wall() - t0;
    }
    void upload_window(int lo, int w, double const *h, int ldh)
    {
        SN_HIP_CHECK(hipMemcpy2DAsync(H + (size_t)lo * ldH + lo, (size_t)ldH * 8, h, (size_t)ldh * 8,
            (size_t)w * 8, w, hipMemcpyHostToDevice, ts));
    }
    void upload_matrix(double *d, double const *h, int ldh, int w)
    {
        SN_HIP_CHECK(hipMemcpy2DAsync(d, (size_t)w * 8, h, (size_t)ldh * 8, (size_t)w * 8, w,
            hipMemcpyHostToDevice, ts));
    }
    // host copies of a window use a padded leading dimension: w*8 bytes is a multiple of 512 for
    // the usual window sizes and the row walks of the sequential kernels would hit a handful
    // of cache sets (measured: 1.5x on the AED kernel at w = 192)
    // (w + 8 is not enough: 184 + 8 = 192 doubles are 24 cache lines, and a stride of 24 lines reaches 8 of
    // the 64 L1 sets -- the AED kernel took 4.6 ms at 184 rows against 3.7 ms at 192.  An ODD number of lines.)
    static int host_ld(int w) { int ld = (w + 8 + 7) / 8 * 8; if ((ld / 8) % 2 == 0) ld += 8; return ld; }

    // small dense Schur problem on a host copy (row S6; schur/cpu.c:402-496)
    int small_block(int lo, int w, double *real, double *imag)
    {
        int const ldh = host_ld(w);
        download_window(lo, w, ws.hWin, ldh);
        for (int j = 0; j < w; j++) for (int i = 0; i < w; i++) ws.hZ[(size_t)j * ldh + i] = (i == j) ? 1.0 : 0.0;
        std::vector<double> wr(w), wi(w);
        int info = host::small_schur(w, ws.hWin, ldh, ws.hZ, ldh, wr.data(), wi.data());
        if (info != 0) return info;
        upload_window(lo, w, ws.hWin, ldh);
        upload_matrix(ws.dZ, ws.hZ, ldh, w);
        apply_transform(lo, w, ws.dZ, w);
        SN_HIP_CHECK(hipStreamSynchronize(ts));     // host buffers are reused
        if (real) for (int i = 0; i < w; i++) { real[lo + i] = wr[i]; imag[lo + i] = wi[i]; }
        st.small_solves++;
        return 0;
    }

    // ---- blocked AED for windows above the hard limit (row S5; schur/core.c:1423-1551
    // perform_large_aed, :1070-1252 perform_deflate_step, :783-1052 perform_deflate_finalize) ------
    // The window is copied into a private matrix and reduced to Schur form RECURSIVELY by this same
    // device path (level 1: chase kernels, MFMA updates, host AED on its small windows).  The
    // deflation checks then run over <= 128-row diagonal windows from the bottom up (host::
    // deflate_window on pinned copies; the rest of the private matrix and of its Schur vectors sees
    // the swaps as in-place MFMA GEMMs); undeflatable blocks are carried along at the top of each
    // window and flushed to the top of the AED window in batches (reorder chains), exactly the
    // reference's scheme.  The spike is embedded as the first column of a padded matrix whose
    // Hessenberg reduction (hessenberg_device) restores the Hessenberg form of the undeflated
    // part; finally the window goes back into H and the accumulated factor reaches H and Q through
    // apply_transform.  Nothing outside the private matrices is touched before the outcome is known.
    void window_updates(LargeAedBuffers &L, int nw, int wb, int w)
    {
        // T(0:wb, win) <- . Zl ; T(win, we:nw) <- Zl^T . ; Z(:, win) <- . Zl   (w <= 128: in place)
        if (wb > 0) dgemm_right_inplace(ts, wb, w, L.dZl, w, L.dT + (size_t)wb * L.ld, L.ld);
        int const rc = nw - (wb + w);
        if (rc > 0) dgemm_left_inplace(ts, w, rc, L.dZl, w, L.dT + (size_t)(wb + w) * L.ld + wb, L.ld);
        dgemm_right_inplace(ts, nw, w, L.dZl, w, L.dZ + (size_t)wb * L.ld, L.ld);
        st.gemm_flops += 2.0 * w * w * ((double)wb + rc + nw);
    }
    void window_to_host(LargeAedBuffers &L, int wb, int w, int ldh)
    {
        SN_HIP_CHECK(hipMemcpy2DAsync(ws.hWin, (size_t)ldh * 8, L.dT + (size_t)wb * L.ld + wb, (size_t)L.ld * 8,
            (size_t)w * 8, w, hipMemcpyDeviceToHost, ts));
        SN_HIP_CHECK(hipStreamSynchronize(ts));
        for (int j = 0; j < w; j++) for (int i = 0; i < w; i++) ws.hZ[(size_t)j * ldh + i] = (i == j) ? 1.0 : 0.0;
    }
    void window_to_device(LargeAedBuffers &L, int wb, int w, int ldh)
    {
        SN_HIP_CHECK(hipMemcpy2DAsync(L.dT + (size_t)wb * L.ld + wb, (size_t)L.ld * 8, ws.hWin, (size_t)ldh * 8,
            (size_t)w * 8, w, hipMemcpyHostToDevice, ts));
        SN_HIP_CHECK(hipMemcpy2DAsync(L.dZl, (size_t)w * 8, ws.hZ, (size_t)ldh * 8, (size_t)w * 8, w,
            hipMemcpyHostToDevice, ts));
    }

    // where a blocked AED spends its time (SN_SCHUR_PROFILE): recursive Schur form of the window, deflation
    // / reordering windows, re-Hessenberg of the undeflated part, the rest (copies, updates of H and Q)
    double prof_laed[4] = {0, 0, 0, 0}; int prof_laed_calls = 0, prof_laed_windows = 0;
    host::AedResult large_aed(int kw, int nw, double sub, double thres, double *spike, double *sr, double *si)
    {
        host::AedResult res{0, 0, 0};
        double const tl0 = wall(); prof_laed_calls++;
        LargeAedBuffers &L = g_large;
        L.ensure(nw);
        int const ld = L.ld;
        constexpr int WD = 128;                 // reorder window (in-place update tiles)
        constexpr int WDD = 96;                 // deflation window: its undeflatable blocks must fit a reorder window with room to move
        copy_matrix(ts, nw, nw, H + (size_t)kw * ldH + kw, ldH, L.dT, ld);
        set_matrix(ts, nw, nw, 0.0, 1.0, L.dZ, ld);
        // (1) Schur form of the window, recursively on the device
        std::vector<double> wr(nw), wi(nw);
        // (the sub-problem inherits the parent's resolved deflation threshold -- u ||H||_F of the WHOLE matrix, the
        // value the deflation checks below use on the same window -- as the reference's does, schur/core.c:1525;
        // 0 stands for the LAPACK criterion)
        SchurParams p1;
        p1.threshold = thres > 0.0 ? thres : -3.0;
        int const rc1 = schur_device(ts, nw, L.dT, ld, L.dZ, ld, wr.data(), wi.data(), p1, nullptr, -1, 1);
        SN_HIP_CHECK(hipStreamSynchronize(ts));
        double const tl1 = wall(); prof_laed[0] += tl1 - tl0;
        if (rc1 != STARNEIG_SUCCESS) {          // no usable Schur form: report the shifts we have, deflate nothing
            res.failed = 1;
            res.shifts = host::order_shifts(nw, wr.data(), wi.data());
            for (int k = 0; k < res.shifts; k++) { sr[k] = wr[k]; si[k] = wi[k]; }
            return res;
        }
        // (2) the spike row sub * Z(0,:) and the block structure
        std::vector<double> sp(nw), tsub(nw, 0.0);
        SN_HIP_CHECK(hipMemcpy2DAsync(sp.data(), 8, L.dZ, (size_t)ld * 8, 8, nw, hipMemcpyDeviceToHost, ts));
        if (nw > 1)
            SN_HIP_CHECK(hipMemcpy2DAsync(tsub.data(), 8, L.dT + 1, (size_t)(ld + 1) * 8, 8, nw - 1, hipMemcpyDeviceToHost, ts));
        SN_HIP_CHECK(hipStreamSynchronize(ts));
        for (int j = 0; j < nw; j++) sp[j] *= sub;
        // (3) deflation windows, bottom up.  [0,top) final undeflatable, [top, bottom-carried)
        // unchecked, [bottom-carried, bottom) carried undeflatable, [bottom, nw) deflated.
        int top = 0, bottom = nw, carried = 0;
        while (top < bottom - carried) {
            int const we = bottom;
            int wb = std::max(top, we - WDD);
            if (wb > top && tsub[wb - 1] != 0.0) wb++;          // do not cut a 2x2 block
            int const w = we - wb, ldh = host_ld(w);
            window_to_host(L, wb, w, ldh); prof_laed_windows++;
            int und = 0;
            int const rej = host::deflate_window(w, ws.hWin, ldh, ws.hZ, ldh, sp.data() + wb, sub, thres, carried, &und);
            window_to_device(L, wb, w, ldh);
            window_updates(L, nw, wb, w);
            SN_HIP_CHECK(hipStreamSynchronize(ts));
            for (int i = 0; i + 1 < w; i++) tsub[wb + i] = ws.hWin[(size_t)i * ldh + i + 1];
            bottom = wb + und; carried = und;
            if (rej) { top = bottom; carried = 0; break; }      // swap rejected: stop testing
            if (wb == top) { top = bottom; carried = 0; break; }
            if (carried >= WDD / 2 || bottom - carried - top < 2) {
                // flush the carried blocks to the top of the AED window (reorder chain)
                int ge = bottom;                                // group = [ge - carried, ge)
                std::vector<int> marks(WD);
                while (ge - carried > top) {
                    int rb = std::max(top, ge - WD);
                    if (rb > top && tsub[rb - 1] != 0.0) rb++;
                    int const rw = ge - rb, rldh = host_ld(rw);
                    window_to_host(L, rb, rw, rldh);
                    for (int i = 0; i < rw; i++) marks[i] = (i >= rw - carried) ? 1 : 0;
                    int failed = 0;
                    int const placed = host::reorder_window(rw, ws.hWin, rldh, ws.hZ, rldh, marks.data(), &failed);
                    // spike segment <- spike * Zl
                    {
                        std::vector<double> t(rw);
                        for (int j = 0; j < rw; j++) { double v = 0.0; for (int k = 0; k < rw; k++) v += sp[rb + k] * ws.hZ[(size_t)j * rldh + k]; t[j] = v; }
                        for (int j = 0; j < rw; j++) sp[rb + j] = t[j];
                    }
                    window_to_device(L, rb, rw, rldh);
                    window_updates(L, nw, rb, rw);
                    SN_HIP_CHECK(hipStreamSynchronize(ts));
                    for (int i = 0; i + 1 < rw; i++) tsub[rb + i] = ws.hWin[(size_t)i * rldh + i + 1];
                    if (failed || placed != carried) {
                        // a rejected swap leaves part of the group behind: everything from here up
                        // counts as undeflatable, the deflation checks end
                        top = bottom; carried = 0; ge = top; break;
                    }
                    ge = rb + carried;
                }
                if (carried > 0) { top += carried; carried = 0; }
            }
        }
        if (carried > 0) { top = bottom; carried = 0; }
        int const ns = top, nd = nw - ns;
        double const tl2 = wall(); prof_laed[1] += tl2 - tl1;
        // shifts: the eigenvalues of the undeflated leading part (all of them if it is tiny)
        {
            std::vector<double> dg(nw), sup(nw, 0.0);
            SN_HIP_CHECK(hipMemcpy2DAsync(dg.data(), 8, L.dT, (size_t)(ld + 1) * 8, 8, nw, hipMemcpyDeviceToHost, ts));
            if (nw > 1)
                SN_HIP_CHECK(hipMemcpy2DAsync(sup.data(), 8, L.dT + ld, (size_t)(ld + 1) * 8, 8, nw - 1, hipMemcpyDeviceToHost, ts));
            SN_HIP_CHECK(hipStreamSynchronize(ts));
            int const cnt = ns >= 2 ? ns : nw;
            for (int i = 0; i < cnt; i++) {
                if (i + 1 < cnt && tsub[i] != 0.0) {
                    double a = dg[i], b = sup[i], c = tsub[i], d = dg[i + 1], cs, sn_;
                    host::lanv2(a, b, c, d, wr[i], wi[i], wr[i + 1], wi[i + 1], cs, sn_);
                    i++;
                } else { wr[i] = dg[i]; wi[i] = 0.0; }
            }
            res.shifts = host::order_shifts(cnt, wr.data(), wi.data());
            for (int k = 0; k < res.shifts; k++) { sr[k] = wr[k]; si[k] = wi[k]; }
        }
        res.deflated = nd;
        if (nd == 0) return res;                // H and Q were never touched
        // (4) spike + Hessenberg form of the undeflated part
        for (int j = 0; j < nw; j++) spike[j] = (j < ns) ? sp[j] : 0.0;
        if (ns > 1 && sub != 0.0) {
            int const np = ns + 1;
            set_matrix(ts, np, np, 0.0, 1.0, L.dQp, ld);
            set_matrix(ts, np, np, 0.0, 1.0, L.dP, ld);
            copy_matrix(ts, ns, ns, L.dT, ld, L.dP + ld + 1, ld);
            SN_HIP_CHECK(hipMemcpyAsync(L.dP + 1, sp.data(), (size_t)ns * 8, hipMemcpyHostToDevice, ts));
            SN_HIP_CHECK(hipStreamSynchronize(ts));     // sp is pageable
            int const pw = 128;         // (the library's default panel width below n = 16000, capi.hip)
            hessenberg_device(ts, np, 0, np, pw, L.dP, ld, L.dQp, ld, nullptr);
            // U = Qp(1:,1:):  T(0:ns, ns:nw) <- U^T . ;  Z(:, 0:ns) <- . U ;  T(0:ns,0:ns) <- P(1:,1:)
            double const *U = L.dQp + ld + 1;
            if (nd > 0) {
                dgemm(ts, 'T', 'N', ns, nd, ns, 1.0, U, ld, L.dT + (size_t)ns * ld, ld, 0.0, L.dTmp, ld);
                copy_matrix(ts, ns, nd, L.dTmp, ld, L.dT + (size_t)ns * ld, ld);
            }
            dgemm(ts, 'N', 'N', nw, ns, ns, 1.0, L.dZ, ld, U, ld, 0.0, L.dTmp, ld);
            copy_matrix(ts, nw, ns, L.dTmp, ld, L.dZ, ld);
            copy_matrix(ts, ns, ns, L.dP + ld + 1, ld, L.dT, ld);
            SN_HIP_CHECK(hipMemcpyAsync(spike, L.dP + 1, 8, hipMemcpyDeviceToHost, ts));
            SN_HIP_CHECK(hipStreamSynchronize(ts));
            st.gemm_flops += 2.0 * ns * ns * ((double)nd + nw);
        }
        double const tl3 = wall(); prof_laed[2] += tl3 - tl2;
        // (5) window back into H, coupling entry, off-window updates of H and Q
        copy_matrix(ts, nw, nw, L.dT, ld, H + (size_t)kw * ldH + kw, ldH);
        if (sub != 0.0)
            hipLaunchKernelGGL(schur_set_entry_kernel, dim3(1), dim3(1), 0, ts,
                H + (size_t)(kw - 1) * ldH + kw, spike[0]);
        apply_transform(kw, nw, L.dZ, ld);
        // the caller reads the deflated diagonal blocks from the host copy of the window
        int const ldh = host_ld(nw);
        SN_HIP_CHECK(hipMemcpy2DAsync(ws.hWin, (size_t)ldh * 8, L.dT, (size_t)ld * 8, (size_t)nw * 8, nw,
            hipMemcpyDeviceToHost, ts));
        prof_laed[3] += wall() - tl3;
        return res;
    }

    // Lazy parts of the window steps: columns [ihi, n) of the left updates, rows [0, T0) of the
    // right updates (each ordered behind the timely updates of its step: entries only ever move
    // from the timely to the lazy zone within a sweep) and Q.  They are issued after the
    // critical path of the sweep so that they execute while the host reduces the AED windows
    // that follow, instead of competing with the sweep for the CUs.
    double prof_scan_wait = 0, prof_dl_wait = 0, prof_issue = 0; int prof_guard_moves = 0;
    double prof_scan_wait_la = 0; int prof_la_cycles = 0, prof_la_done_at_chain_end = 0; double prof_la_chain_s = 0;
    double prof_la_t0 = 0; hipEvent_t prof_la_ev = nullptr;
    struct LazyItem { SweepStep step; int ev; int row_split; };
    std::vector<LazyItem> lazy;
    void launch_q(std::vector<LazyItem> const &items)
    {
        for (LazyItem const &it : items) {
            double *Ubuf = ws.dU + (size_t)it.ev * ws.max_chains * WS_MAX * WS_MAX;
            // (64-row tiles reach 13 % more of the HBM rate alone, scratch/update_bench.py, but change nothing in situ)
            hipLaunchKernelGGL(schur_update_kernel<3>, dim3(divceil(nq, 128), it.step.ntasks),
                dim3(256), UPDATE_LDS_BYTES_R, ws.qs, it.step, H, ldH, Q, ldQ, nq, Ubuf, 0, nq);
        }
    }

    void flush_lazy(int col_split)
    {
        if (lazy.empty()) return;
        // the far stream is in order: the timely updates of the last step cover all earlier ones
        int const last_ev = lazy.back().ev;
        int const fslot = (int)(ws.flush_total % SchurWorkspace::FLUSH_RING);
        SN_HIP_CHECK(hipStreamWaitEvent(ws.hs, ws.far_done[last_ev], 0));
        if (Q) SN_HIP_CHECK(hipStreamWaitEvent(ws.qs, ws.near_done[last_ev], 0));
        for (LazyItem const &it : lazy) {
            int const ntasks = it.step.ntasks;
            double *Ubuf = ws.dU + (size_t)it.ev * ws.max_chains * WS_MAX * WS_MAX;
            int const lazy_cols = n - col_split, lazy_rows = it.row_split;
            if (shard_world > 1 && lazy_cols > 0) {
                // deflated column tiles have one owner each (schur_update_pair_sharded_kernel)
                int const first_own = divceil(it.step.ihi, 128);
                int const rep_tiles = std::max(0, divceil(first_own * 128 - col_split, 128));
                int const own_tiles = std::max(0, divceil(divceil(n, 128) - first_own, shard_world));
                hipLaunchKernelGGL(schur_update_pair_sharded_kernel,
                    dim3(std::max(rep_tiles + own_tiles, divceil(std::max(lazy_rows, 1), 128)), ntasks, lazy_rows > 0 ? 2 : 1),
                    dim3(256), UPDATE_LDS_BYTES_P, ws.hs, it.step, H, ldH, n, Ubuf, col_split, n, 0, lazy_rows,
                    rep_tiles, shard_rank, shard_world);
            } else if (lazy_cols > 0 || lazy_rows > 0)
                hipLaunchKernelGGL(schur_update_pair_kernel,
                    dim3(divceil(std::max(lazy_cols, lazy_rows), 128), ntasks, lazy_rows > 0 ? 2 : 1), dim3(256),
                    UPDATE_LDS_BYTES_P, ws.hs, it.step, H, ldH, n, Ubuf, col_split, n, 0, lazy_rows);
            ws.slot_flush[it.ev] = ws.flush_total;
        }
        if (Q) launch_q(lazy);
        SN_HIP_CHECK(hipEventRecord(ws.h_done[fslot], ws.hs));
        if (Q) SN_HIP_CHECK(hipEventRecord(ws.q_done[fslot], ws.qs));
        ws.flush_total++;
        lazy.clear();
    }

    // ---- one multi-shift sweep over the active block [ilo, ihi), issued in up to two phases ----
    // sweep_begin fixes the chains; sweep_issue(limit) issues window steps while the leading
    // chain (window + near strip) stays above row `limit`; sweep_finish() runs the rest down to
    // the (possibly smaller) ihi of that moment.  Phase A / phase B of the look-ahead scheme in
    // schur_device: the head of a sweep runs while the host still reduces AED windows at the
    // bottom of the block.
    struct SweepState {
        bool active = false;
        int ilo = 0, ihi = 0, ws_ = 0, nbc = 0, adv = 0, gap = 1, nbulges = 0, chains = 0;
        int t = 0, last_t = -2;
        long issued = 0, last_waited_flush = -1;
        int col_split = 0;          // columns >= col_split are lazy for the steps being issued
    } sw;
    double sweep_flops = 0.0; int sweep_launches = 0;
    int spw_cap = -1;           // conf->shifts_per_window (process_args.c:418-437)
    long chain_passes = 0;      // chains over all sweeps: the rounding error grows like its square root
    int lazy_batch = tuning().schur_lazy_batch;

    void sweep_begin(int ilo, int ihi, int nshifts, double const *sr, double const *si)
    {
        int const size = ihi - ilo;
        int nbulges = nshifts / 2;
        int ws_ = std::min(WS_MAX, size);
        int nbc = std::min(NB_MAX, (ws_ - 1) / 6);          // so that 2*(3 nbc) + 1 <= ws
        if (spw_cap > 0) nbc = std::min(nbc, spw_cap / 2);  // conf->shifts_per_window
        if (nbc < 1) nbc = 1;
        if (size <= WS_MAX) nbc = std::min(nbulges, std::max(1, (size - 1) / 3));
        nbc = std::min(nbc, NB_MAX);
        nbulges = std::min(nbulges, nbc * ws.max_chains);       // U buffers hold max_chains windows per step
        sw = SweepState{};
        sw.active = true;
        sw.ilo = ilo; sw.ihi = ihi; sw.ws_ = ws_; sw.nbc = nbc; sw.nbulges = nbulges;
        sw.chains = divceil(nbulges, nbc);
        chain_passes += sw.chains;
        sw.adv = ws_ - 1 - 3 * nbc;                           // columns a chain advances per step
        // chains ws+adv rows apart: a chain's next window then depends on its OWN near update only
        sw.gap = (sw.adv > 0) ? divceil(ws_ + sw.adv, sw.adv) : 1;
        sw.col_split = ihi;
        // the caller's arrays are reused by the next AED at once: stage the shifts in pinned memory
        // (two alternating buffers: the previous sweep's upload is long done when its buffer returns)
        double *stage = ws.hShift + (size_t)(ws.shift_uploads++ & 1) * 2 * 8 * ws.nwmax;
        std::memcpy(stage, sr, (size_t)nshifts * 8);
        std::memcpy(stage + 8 * ws.nwmax, si, (size_t)nshifts * 8);
        SN_HIP_CHECK(hipMemcpyAsync(ws.dShiftR, stage, (size_t)nshifts * 8, hipMemcpyHostToDevice, s));
        SN_HIP_CHECK(hipMemcpyAsync(ws.dShiftI, stage + 8 * ws.nwmax, (size_t)nshifts * 8, hipMemcpyHostToDevice, s));
        // At the start of a sweep the chains are back at the top: rows that were above the band
        // (lazy) in the previous sweep are in the band (timely) again.
        wait_lazy_h();
    }

    // positions of a chain: p = 0 introduce at ilo, then lo = ilo + p*adv until the window
    // reaches ihi (finalize).  Single-window blocks run in FULL mode.
    int steps_per_chain() const
    {
        int const size = sw.ihi - sw.ilo;
        return (size <= WS_MAX) ? 1 : divceil(size - sw.ws_, sw.adv) + 1;
    }

    // Pipeline: critical stream s:  chase(t) -> [wait far(t-1)] near(t)
    //           far stream f     :  [wait near(t)] timely far-left(t), timely right(t)
    //           lazy streams     :  the lazy parts of far-left / right;  Q  (flush_lazy)
    // so chase(t+1) overlaps far(t).  The U factors of a step live in a ring slot until the
    // lazy streams have consumed them.  Returns true when the sweep is complete.
    bool sweep_issue(int limit)
    {
        int const ilo = sw.ilo, ihi = sw.ihi, ws_ = sw.ws_, adv = sw.adv, gap = sw.gap, chains = sw.chains;
        int const spc = steps_per_chain();
        int const total_steps = spc + (chains - 1) * gap;
        SweepStep step{ilo, ihi, ws_, sw.nbc, adv, gap, sw.nbulges, spc, 0, 0, 0};
        hipStream_t const f = ws.far;
        int const col_split = sw.col_split;
        for (; sw.t < total_steps; sw.t++) {
            int const t = sw.t;
            int cmin = (t - spc + 1 + gap - 1) / gap;      // ceil for positives
            if (t - spc + 1 <= 0) cmin = 0;
            int const cmax = std::min(chains - 1, t / gap);
            if (cmax < cmin) continue;
            // the leading chain is the first one still in flight
            if (limit < ihi && ilo + (t - cmin * gap) * adv + ws_ + adv > limit) return false;
            step.t = t; step.cmin = cmin; step.ntasks = cmax - cmin + 1;
            int const ntasks = step.ntasks;
            // U buffers and events live in a ring indexed by a counter that runs across sweeps
            int const ev = (int)(ws.issued_total % ws.ring);
            int const evp = (int)((ws.issued_total + ws.ring - 1) % ws.ring);
            double *Ubuf = ws.dU + (size_t)ev * ws.max_chains * WS_MAX * WS_MAX;
            if (ws.slot_flush[ev] >= 0) {
                // the lazy streams must be through with the previous tenant of this U slot
                long const fid = ws.slot_flush[ev];
                if (fid != sw.last_waited_flush) {
                    if (ws.flush_total - fid < SchurWorkspace::FLUSH_RING) {
                        SN_HIP_CHECK(hipStreamWaitEvent(s, ws.h_done[(int)(fid % SchurWorkspace::FLUSH_RING)], 0));
                        if (Q) SN_HIP_CHECK(hipStreamWaitEvent(s, ws.q_done[(int)(fid % SchurWorkspace::FLUSH_RING)], 0));
                    } else {        // the batch events were recycled: wait for everything issued so far
                        wait_lazy_h();
                        if (Q) {
                            SN_HIP_CHECK(hipEventRecord(ws.lazy_mark, ws.qs));
                            SN_HIP_CHECK(hipStreamWaitEvent(s, ws.lazy_mark, 0));
                        }
                    }
                    sw.last_waited_flush = fid;
                }
                ws.slot_flush[ev] = -1;
            }
            // chase(t) may overlap far(t-1) only in the regular regime.  On short active blocks
            // steps are skipped (a chain finishes before the next one is introduced) and the new
            // chain's first window would race with the finished chain's pending updates
            // (tests/test_schur_pipeline.py checks this rule on a model of the schedule).
            if (sw.issued > 0 && sw.last_t != t - 1) SN_HIP_CHECK(hipStreamWaitEvent(s, ws.far_done[evp], 0));
            hipLaunchKernelGGL(schur_chase_ulds_kernel, dim3(ntasks), dim3(CHASE_THREADS), CHASE_LDS_BYTES_WU, s,
                step, H, ldH, Ubuf, ws.dShiftR, ws.dShiftI);
            sweep_launches++;
            int min_lo = n, max_lo = 0;
            for (int k = 0; k < ntasks; k++) {
                ChaseTask const tk = make_task(step, k);
                int const rc = n - (tk.lo + tk.n);
                min_lo = std::min(min_lo, tk.lo); max_lo = std::max(max_lo, tk.lo);
                sweep_flops += 2.0 * tk.n * tk.n * ((double)rc + tk.lo + (Q ? nq : 0));
            }
            // rows above T0 are out of reach of every chain still in flight and of the AED windows
            // that follow this sweep (guard row)
            // (while chains are still being introduced at the top, every row is within reach)
            int const rear = (cmax == chains - 1) ? min_lo : ilo;
            int const row_split = std::max(0, std::min(rear, ws.guard_row));
            if (sw.issued > 0) SN_HIP_CHECK(hipStreamWaitEvent(s, ws.far_done[evp], 0));
            hipLaunchKernelGGL(schur_update_kernel<2>, dim3(1, ntasks), dim3(256),
                UPDATE_LDS_BYTES_L, s, step, H, ldH, Q, ldQ, n, Ubuf, 0, n);
            SN_HIP_CHECK(hipEventRecord(ws.near_done[ev], s));
            SN_HIP_CHECK(hipStreamWaitEvent(f, ws.near_done[ev], 0));
            // timely: columns [lo+w+adv, col_split) and rows [T0, lo).  Two launches: the left update of
            // one chain and the right update of a chain ahead of it meet in the same entries
            int const timely_cols = col_split - (min_lo + ws_ + adv), timely_rows = max_lo - row_split;
            if (timely_cols > 0)
                hipLaunchKernelGGL(schur_update_kernel<0>, dim3(divceil(timely_cols, 128), ntasks), dim3(256),
                    UPDATE_LDS_BYTES_L, f, step, H, ldH, Q, ldQ, n, Ubuf, 0, col_split);
            if (timely_rows > 0)
                hipLaunchKernelGGL(schur_update_kernel<1>, dim3(divceil(timely_rows, 128), ntasks), dim3(256),
                    UPDATE_LDS_BYTES_R, f, step, H, ldH, Q, ldQ, n, Ubuf, row_split, n);
            SN_HIP_CHECK(hipEventRecord(ws.far_done[ev], f));
            // the lazy parts of this step are issued later (flush_lazy)
            lazy.push_back(LazyItem{step, ev, row_split});
            // Phase A of a look-ahead sweep is off the host's critical path and latency-bound on
            // the GPU: its lazy updates are issued in small batches and fill the idle CUs, so that
            // they are through when the AED chain ends.  Otherwise they wait for the sweep's end.
            int const batch = (limit < ihi) ? lazy_batch : ws.ring / 2;
            if ((int)lazy.size() >= batch) flush_lazy(col_split);
            sw.issued++;
            ws.issued_total++;
            sw.last_t = t;
        }
        return true;
    }

    void sweep_finish()
    {
        sweep_issue(sw.ihi);
        // the critical part of the sweep is complete when the far stream has drained
        if (sw.issued > 0)
            SN_HIP_CHECK(hipStreamWaitEvent(s, ws.far_done[(int)((ws.issued_total - 1) % ws.ring)], 0));
        flush_lazy(sw.col_split);
        sw.active = false;
        st.sweeps++;
        st.gemm_flops += sweep_flops; st.chase_launches += sweep_launches;
        sweep_flops = 0.0; sweep_launches = 0;
    }

    void sweep(int ilo, int ihi, int nshifts, double const *sr, double const *si)
    {
        sweep_begin(ilo, ihi, nshifts, sr, si);
        sweep_finish();
    }
};

} // namespace

int schur_device(hipStream_t caller, int n, double *dH, int ldH, double *dQ, int ldQ,
    double *real, double *imag, SchurParams const &prm, SchurStats *stats, int q_rows, int level)
{
    // the whole reduction runs on the library's own stream pair (never on the legacy NULL
    // stream), fenced against the caller's stream at entry and exit
    hipStream_t &own = own_[level];
    hipEvent_t &fence = fence_[level];
    if (!own) {
        // The critical stream gets the highest priority.  Besides the scheduling preference this
        // puts it into another pool of hardware queues than the far and lazy streams: the runtime
        // multiplexes streams of one priority onto a few hardware queues, and with other streams
        // around (an RCCL communicator costs the reduction 0.8 s otherwise) the chase kernels would
        // queue behind lazy updates.
        int lo_prio = 0, hi_prio = 0;
        SN_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio));
        make_stream(&own, true, hi_prio);
        SN_HIP_CHECK(hipEventCreateWithFlags(&fence, hipEventDisableTiming));
    }
    hipStream_t s = own;
    SN_HIP_CHECK(hipEventRecord(fence, caller));
    SN_HIP_CHECK(hipStreamWaitEvent(s, fence, 0));

    // ---- parameters (schur/process_args.c:116-162, :271-288, :356) --------------------------
    int const min_val = lapack_min_shifts(n);
    int nw_default = (int)std::max(min_val / 0.7, 0.08 * n);
    int ns_default = (int)std::max((double)min_val, 0.06 * n);
    // The AED window is reduced on the host (O(w^3) work per call, one chain of small reflectors)
    // and every shift pair drives several bulges (see `reuse` below), so the best window is much
    // smaller than the reference's 0.08 n: a larger one saves sweeps and costs host time.  With
    // the helper team of the window kernel (schur_host_team.h) the optimum at n = 20000 is flat
    // between 224 / 140 and 288 / 180 (1.82-1.92 s; 2.3 s at 160 / 106), without it around
    // 192 / 128 - 208 / 128 (2.15 s); explicit conf values are honoured up to 1024.
    int const helpers = level == 0 ? (tuning().schur_helpers >= 0 ? std::min(tuning().schur_helpers, prm.host_threads - 1)
                                                                 : (prm.host_threads >= 6 ? 5 : 0)) : 0;
    // The balance moves with n (the sweeps get longer, the window kernel does not): 0.21 s at 192 / 128
    // against 0.31 s at 256 / 160 for n = 4000, 0.46 s against 0.58 s for n = 8000.
    // (a replica that carries a row block of Q -- the sharded Schur leg -- must choose the same window as
    // its peers: the choice depends on the `cores` the caller states, never on how many helper threads
    // are actually running, and the callers give every rank the same `cores` -- node_team.hip by
    // construction, distributed.py checks it; the helper team itself is bit-identical to the serial kernel)
    int const nw_cap = std::min(helpers >= 2 ? 288 : 192, (136 + (int)(0.006 * n) + 8) / 16 * 16);
    nw_default = std::min(nw_default, nw_cap);
    ns_default = std::min(ns_default, nw_cap * 5 / 8);
    // AED windows above the hard limit (process_args.c:372-398, default 300) are reduced by the
    // blocked device path (Driver::large_aed, row S5), the others by the sequential host kernel;
    // the private window of a blocked AED (level 1) always takes the small defaults
    int const hard_limit = prm.aed_parallel_hard_limit > 0 ? prm.aed_parallel_hard_limit : 300;
    int nw_conf = prm.aed_window_size > 0 ? prm.aed_window_size : nw_default;
    if (level > 0) nw_conf = std::min(nw_conf, std::min(hard_limit, 300));
    // a replica that carries a row block of Q (q_rows < n: the sharded Schur leg) must stay
    // bit-identical to its peers, and the blocked device AED is not (atomics in its
    // re-Hessenberg step): such a call keeps its windows on the sequential host kernel
    if (q_rows >= 0 && q_rows < n) nw_conf = std::min(nw_conf, hard_limit);
    nw_conf = std::min(nw_conf, std::max(4, n));
    int ns_conf = prm.shift_count > 0 ? prm.shift_count : ns_default;
    ns_conf = std::min(ns_conf, 9 * nw_conf / 10);
    ns_conf = std::max(2, ns_conf - ns_conf % 2);
    int const small_limit = prm.small_limit > 0 ? std::min(prm.small_limit, 1024)
                                                : std::max(WS_MAX + 32, std::min(300, nw_conf));
    int const nibble = prm.aed_nibble > 0 ? prm.aed_nibble : 40;
    int const iter_limit = prm.iteration_limit > 0 ? prm.iteration_limit : 300;

    SchurWorkspace &ws = g_sws[level];
    int const wmax = std::max({nw_conf, small_limit, 2 * WS_MAX});
    // Shift multiplicity: every shift pair of an AED drives `reuse` bulges of the following
    // sweep (the AED window bounds the number of distinct shifts, the host AED kernel bounds the
    // window).  A sweep is latency-bound by its first chain (steps_per_chain window steps), more
    // chains only add `gap` steps each, so the same shifts applied 8x cost ~1.3x the time of a
    // sweep and cut the number of sweeps at n = 20000 from 81 to 13 (measured: 7.0 s -> 5.4 s at
    // a 192-row AED window).
    // Rounds 1-3 let the multiplicity grow with the size (2 below n = 4000, 4 below 12000, 8 above); measured
    // again in round 4 (default / 8): n = 2000 0.16 / 0.14 s, n = 4000 0.24 / 0.22 s, n = 8000 0.51 / 0.45 s at
    // 55 / 59 u -- 8 from n = 1000 on.  SN_SCHUR_REUSE=k overrides.
    int const reuse_env = tuning().schur_reuse;
    // (a conf with many shifts -- the reference's 0.06 n -- fills the sweep by itself: the multiplicity is
    // capped so that a sweep carries ~450 bulges, what 8 x 53 give at the default sizes)
    int const reuse_n = n < 1000 ? 2 : 8;
    int const reuse = reuse_env ? reuse_env : std::max(1, std::min(reuse_n, 450 / std::max(1, ns_conf / 2)));
    ws.ensure(n, wmax, divceil(reuse * (ns_conf / 2), NB_MAX) + 2);
    ws.guard_row = 0;
    Driver d{s, n, dH, ldH, dQ, ldQ, ws, SchurStats{}};

    hipEvent_t e0, e1;
    SN_HIP_CHECK(hipEventCreate(&e0)); SN_HIP_CHECK(hipEventCreate(&e1));
    SN_HIP_CHECK(hipEventRecord(e0, s));

    // ---- deflation threshold: u * ||H||_F by default (schur/core.c:2390-2436) --------------
    double thres = prm.threshold;
    if (thres == -1.0 || thres == -2.0) {
        double h = 0.0;
        // (fixed summation order: every replica of the reduction gets the same threshold bit for bit)
        sumsq_ordered(s, n, n, dH, ldH, ws.dTmp, ws.dAcc);
        SN_HIP_CHECK(hipMemcpyAsync(&h, ws.dAcc, 8, hipMemcpyDeviceToHost, s));
        SN_HIP_CHECK(hipStreamSynchronize(s));
        thres = DBL_EPSILON * std::sqrt(h);
    } else if (thres == -3.0) thres = 0.0;      // LAPACK-style criteria
    else if (thres < 0.0) return STARNEIG_INVALID_CONFIGURATION;

    // The helper team of the host window kernel (schur_host_team.h) for the duration of this reduction.
    // Opened only now: for the session the calling thread is pinned to one core, and a thread that the
    // runtime creates in the meantime would inherit that -- the workspace, its streams and the first
    // synchronisation are behind us here.
    struct HelperSession {
        bool on;
        explicit HelperSession(int count) : on(count >= 2) { if (on) host::helper_session(true, count); }
        ~HelperSession() { if (on) host::helper_session(false); }
    } helper_session(helpers);

    std::vector<double> sr(8 * wmax), si(8 * wmax), spike(wmax);
    // Look-ahead: after a sweep the next one is started at once with the shifts at hand (stale
    // by one AED chain -- measured: no effect on the number of sweeps) and runs down to the
    // guard row while the host reduces the AED windows below it on a separate stream; when the
    // chain of AEDs ends, the sweep continues through the rest of the block (phase B).
    bool const lookahead = !tuning().schur_nolookahead;
    std::vector<double> stale_r, stale_i;
    d.ts = s;
    d.nq = (q_rows >= 0) ? q_rows : n;
    if (level == 0 && prm.shard_world > 1) { d.shard_rank = prm.shard_rank; d.shard_world = prm.shard_world; }
    d.spw_cap = prm.shifts_per_window;
    // (Round 2 switched the multiplicity off on slowly converging inputs -- fewer than 5 AEDs per sweep
    // over four sweeps -- because the extra chain passes cost accuracy there: all-ones Hessenberg,
    // n = 8000, 1205 u at a multiplicity of 4 against 737 u at 1.  Since the reflectors are scaled by
    // exact powers of two the same matrices lose 10-30 u (all-ones 165 -> 176 u, Toeplitz 177 -> 207 u,
    // companion 391 -> 388 u) and gain a third of the time (1.26 -> 0.80 s); and with the larger AED
    // windows of round 3 the rule misfired on random matrices whose chains are short in AEDs because each
    // AED deflates more (n = 8000 at 184 rows: 29 sweeps instead of 10).  It is gone.)
    int aeds_since_sweep = 0;
    auto replicate = [&](int nsh) {
        aeds_since_sweep = 0;
        for (int r = 1; r < reuse; r++)
            for (int k = 0; k < nsh; k++) { sr[r * nsh + k] = sr[k]; si[r * nsh + k] = si[k]; }
        return nsh * reuse;
    };
    auto finish_lookahead = [&](int ihi_now) {
        if (tuning().schur_profile && d.prof_la_ev && d.prof_la_t0 > 0) {
            d.prof_la_chain_s += wall() - d.prof_la_t0; d.prof_la_t0 = 0;
            if (hipEventQuery(d.prof_la_ev) == hipSuccess) d.prof_la_done_at_chain_end++;
        }
        // phase B: the AED stream is done with the bottom of the block, the lazy H updates issued
        // so far (phase A, AED) are waited for -- rows near the guard row become timely again
        SN_HIP_CHECK(hipEventRecord(ws.aed_mark, ws.aed));
        SN_HIP_CHECK(hipStreamWaitEvent(s, ws.aed_mark, 0));
        d.ts = s;
        // the next AED chain will work above ihi_now - 8 nw: make those rows timely now, while
        // the lazy stream is waited for anyway
        ws.guard_row = std::max(0, std::min(ws.guard_row, ihi_now - 8 * nw_conf));
        d.wait_lazy_h();
        d.sw.ihi = ihi_now; d.sw.col_split = ihi_now;
        double const t_issue = wall();
        d.sweep_finish();
        d.prof_issue += wall() - t_issue;
    };
    int rc = STARNEIG_SUCCESS;
    int ihi = n;                    // H(ihi:n, ihi:n) is already quasi-triangular
    int iter = 0, stagnation = 0;
    while (ihi > 0) {
        // ---- locate the active block [ilo, ihi) ------------------------------------------------
        bool const la = d.sw.active;        // the head of a sweep is in flight above the guard row
        int const scan_lo = la ? ws.guard_row : 0;
        if (la && ihi - scan_lo < 4) { finish_lookahead(ihi); continue; }
        hipLaunchKernelGGL(schur_scan_subdiag_kernel, dim3(divceil(std::max(ihi - scan_lo - 1, 1), 256)), dim3(256),
            0, d.ts, scan_lo, ihi, dH, ldH, thres, ws.dSub, n);
        if (ihi - scan_lo > 1) {
            double tw = wall();
            SN_HIP_CHECK(hipMemcpyAsync(ws.hSub + scan_lo, ws.dSub + scan_lo, (size_t)(ihi - 1 - scan_lo) * 8,
                hipMemcpyDeviceToHost, d.ts));
            SN_HIP_CHECK(hipStreamSynchronize(d.ts));
            d.st.wait_s += wall() - tw; d.prof_scan_wait += wall() - tw;
            if (la) d.prof_scan_wait_la += wall() - tw;
        }
        int ilo = ihi - 1;
        while (ilo > scan_lo && ws.hSub[ilo - 1] != 0.0) ilo--;
        // during look-ahead ilo == guard row means: no split found below the guard, the block
        // continues upwards to the top of the sweep in flight
        bool const open_top = la && ilo == scan_lo;
        int const size = open_top ? ihi - d.sw.ilo : ihi - ilo;
        if (size == 1) {
            if (real) {
                double v; SN_HIP_CHECK(hipMemcpy(&v, dH + (size_t)ilo * ldH + ilo, 8, hipMemcpyDeviceToHost));
                real[ilo] = v; imag[ilo] = 0.0;
            }
            ihi = ilo; continue;
        }
        if (size <= small_limit) {
            // (a sweep head in flight above an open block: its bulges sit inside the rows this
            // solve would touch and `ilo` is only the guard row -- let the sweep through first)
            if (la && open_top) { finish_lookahead(ihi); continue; }
            if (ilo < ws.guard_row) d.set_guard_row(ilo);
            int info = d.small_block(ilo, size, real, imag);
            if (info != 0) { rc = STARNEIG_DID_NOT_CONVERGE; break; }
            ihi = ilo; stagnation = 0; continue;
        }
        if (iter >= iter_limit * std::max(1, n / std::max(1, ns_conf))) { rc = STARNEIG_DID_NOT_CONVERGE; break; }

        // ---- start the next sweep ahead of the AED chain -------------------------------------------
        if (!la && lookahead && !stale_r.empty() && stagnation == 0) {
            int const r1 = ihi - 8 * nw_conf;
            if (r1 - ilo >= 4 * (WS_MAX + 64)) {
                int nsh = (int)stale_r.size();
                for (int k = 0; k < nsh; k++) { sr[k] = stale_r[k]; si[k] = stale_i[k]; }
                nsh = replicate(nsh);
                // (the guard row was already lowered to this value before phase B of the previous
                // sweep, so that nothing lazy is pending below it and no wait is needed here)
                d.set_guard_row(r1);
                // the AED stream starts where the critical stream stands now: behind the previous
                // sweep and behind the lazy updates of rows that a lowered guard row made timely
                SN_HIP_CHECK(hipEventRecord(ws.aed_mark, s));
                SN_HIP_CHECK(hipStreamWaitEvent(ws.aed, ws.aed_mark, 0));
                double const t_issue = wall();
                d.sweep_begin(ilo, ihi, nsh, sr.data(), si.data());
                d.sw.col_split = r1;            // phase A: the columns of the AED region are lazy
                d.sweep_issue(r1);
                d.flush_lazy(r1);               // ... run while the host is busy with the AEDs
                d.prof_issue += wall() - t_issue;
                if (tuning().schur_profile) {   // is the head of the sweep through when the AED chain ends?
                    if (!d.prof_la_ev) SN_HIP_CHECK(hipEventCreateWithFlags(&d.prof_la_ev, hipEventDisableTiming));
                    SN_HIP_CHECK(hipEventRecord(d.prof_la_ev, s));
                    d.prof_la_t0 = wall(); d.prof_la_cycles++;
                }
                d.ts = ws.aed;
                iter++;
            }
        }
        bool const la_now = d.sw.active;

        // ---- aggressive early deflation on the trailing window ---------------------------------
        int nw = std::min(nw_conf, size);
        if (stagnation > 0) nw = std::min(size, std::min(wmax, nw + nw / 20 * stagnation + 2));  // core.c:1912-1918
        if (la_now && ihi - nw <= ws.guard_row + 1) {
            // the AED chain has reached the guard row: let the sweep through first
            finish_lookahead(ihi);
            continue;
        }
        int const kw = ihi - nw;
        double sub = 0.0;
        if (kw > ilo || (open_top && kw > d.sw.ilo)) sub = ws.hSub[kw - 1];
        int const ldh = Driver::host_ld(nw);
        if (!la_now && kw < ws.guard_row) d.set_guard_row(kw - 4 * nw);
        host::AedResult ar;
        bool const blocked = level == 0 && nw > hard_limit;
        double t_aed0 = wall();
        if (blocked) {
            ar = d.large_aed(kw, nw, sub, thres, spike.data(), sr.data(), si.data());
            d.st.aed_host_s += wall() - t_aed0;
        } else {
            d.download_window(kw, nw, ws.hWin, ldh);
            t_aed0 = wall();
            ar = host::aed_window(nw, ws.hWin, ldh, ws.hZ, ldh, sub, thres,
                spike.data(), sr.data(), si.data());
            d.st.aed_host_s += wall() - t_aed0;
        }
        d.st.aeds++;
        aeds_since_sweep++;
        if (ar.deflated > 0) {
            if (!blocked) {
                d.upload_window(kw, nw, ws.hWin, ldh);
                d.upload_matrix(ws.dZ, ws.hZ, ldh, nw);
                if (sub != 0.0)
                    hipLaunchKernelGGL(schur_set_entry_kernel, dim3(1), dim3(1), 0, d.ts,
                        dH + (size_t)(kw - 1) * ldH + kw, spike[0]);
                d.apply_transform(kw, nw, ws.dZ, nw);
            }
            SN_HIP_CHECK(hipStreamSynchronize(d.ts));
            if (real) {
                std::vector<double> wr(nw), wi(nw);
                host::extract_eigenvalues(ar.deflated, ws.hWin + (size_t)(nw - ar.deflated) * ldh + (nw - ar.deflated),
                    ldh, wr.data(), wi.data());
                for (int i = 0; i < ar.deflated; i++) {
                    real[ihi - ar.deflated + i] = wr[i]; imag[ihi - ar.deflated + i] = wi[i];
                }
            }
            ihi -= ar.deflated;
            stagnation = 0;
        } else stagnation++;
        int const size_now = open_top ? ihi - d.sw.ilo : ihi - ilo;
        if (size_now <= small_limit && !la_now) continue;
        // enough deflation: try AED again before spending a sweep (nibble rule, process_args.c:356)
        if (100 * ar.deflated > nibble * nw && size_now > small_limit) continue;
        int nshifts = std::min(ar.shifts, ns_conf);
        nshifts -= nshifts % 2;
        // Exceptional shifts (the LAPACK dlaqr0 recipe, every 6th sweep without deflation, or
        // when the window offers no usable shift -- e.g. a nilpotent trailing block): pairs
        // from [0.75 s + h_ii, s; -0.4375 s, 0.75 s + h_ii], s = |h_{i,i-1}| + |h_{i-1,i-2}|.
        if (!la_now && (nshifts < 2 || (stagnation > 0 && stagnation % 6 == 0))) {
            int const want = std::max(2, std::min(ns_conf, (ihi - ilo - 2) / 2 * 2));
            std::vector<double> dg(want + 2);
            int const first = ihi - (want + 2) >= 0 ? ihi - (want + 2) : 0;
            int const cnt = ihi - first;
            SN_HIP_CHECK(hipMemcpy2DAsync(dg.data(), 8, dH + (size_t)first * ldH + first, (size_t)(ldH + 1) * 8,
                8, cnt, hipMemcpyDeviceToHost, s));
            SN_HIP_CHECK(hipStreamSynchronize(s));
            nshifts = 0;
            for (int i = ihi - 1; i >= ilo + 2 && nshifts + 2 <= want; i -= 2) {
                double ss = std::fabs(ws.hSub[i - 1]) + std::fabs(ws.hSub[i - 2]);
                double aa = 0.75 * ss + dg[i - first], bb = ss, cc = -0.4375 * ss, dd = aa, cs, sn;
                host::lanv2(aa, bb, cc, dd, sr[nshifts], si[nshifts], sr[nshifts + 1], si[nshifts + 1], cs, sn);
                if (ss == 0.0) { sr[nshifts] = sr[nshifts + 1] = dg[i - first] + 1e-3 * (1.0 + std::fabs(dg[i - first])); si[nshifts] = si[nshifts + 1] = 0.0; }
                nshifts += 2;
            }
            if (nshifts < 2) { rc = STARNEIG_DID_NOT_CONVERGE; break; }
        }
        if (stagnation > 60) { rc = STARNEIG_DID_NOT_CONVERGE; break; }
        // the shifts of this AED serve the sweep that is started after the one in flight
        if (nshifts >= 2) { stale_r.assign(sr.begin(), sr.begin() + nshifts); stale_i.assign(si.begin(), si.begin() + nshifts); }

        // ---- multi-shift sweep -----------------------------------------------------------------------
        if (la_now) { finish_lookahead(ihi); continue; }
        if (nshifts < 2) { rc = STARNEIG_DID_NOT_CONVERGE; break; }
        if (reuse > 1 && ihi - ilo > 4 * WS_MAX) nshifts = replicate(nshifts);
        // rows below the guard row stay timely: room for the AED windows that follow the sweep
        d.set_guard_row(ihi - 8 * nw_conf);
        double const t_issue = wall();
        d.sweep(ilo, ihi, nshifts, sr.data(), si.data());
        d.prof_issue += wall() - t_issue;
        iter++;
    }
    if (d.sw.active) finish_lookahead(ihi);     // (error exits: no bulges are left behind)
    // the lazy streams have to drain before the result is complete
    SN_HIP_CHECK(hipEventRecord(ws.lazy_mark, ws.hs));
    SN_HIP_CHECK(hipStreamWaitEvent(s, ws.lazy_mark, 0));
    SN_HIP_CHECK(hipEventRecord(fence, ws.qs));
    SN_HIP_CHECK(hipStreamWaitEvent(s, fence, 0));
    SN_HIP_CHECK(hipEventRecord(e1, s));
    SN_HIP_CHECK(hipStreamWaitEvent(caller, e1, 0));
    SN_HIP_CHECK(hipEventSynchronize(e1));
    SN_HIP_CHECK(hipEventElapsedTime(&d.st.total_ms, e0, e1));
    SN_HIP_CHECK(hipEventDestroy(e0)); SN_HIP_CHECK(hipEventDestroy(e1));
    if (tuning().schur_profile && d.prof_laed_calls > 0)
        fprintf(stderr, "[schur] blocked AED: %d calls, %d deflation windows: Schur form of the window %.3f s, deflation / reordering %.3f s, "
            "re-Hessenberg %.3f s, write-back and updates (issue) %.3f s\n", d.prof_laed_calls, d.prof_laed_windows,
            d.prof_laed[0], d.prof_laed[1], d.prof_laed[2], d.prof_laed[3]);
    if (tuning().schur_profile)
        fprintf(stderr, "[schur] look-ahead: %d sweeps started ahead of their AED chain; the head was through when the chain ended in %d; "
            "host time from the head's issue to the chain's end %.3f s; scan-sync wait while a head is in flight %.3f s\n",
            d.prof_la_cycles, d.prof_la_done_at_chain_end, d.prof_la_chain_s, d.prof_scan_wait_la);
    if (tuning().schur_profile)
        fprintf(stderr, "[schur] total %.3f s: aed_host %.3f, scan-sync wait %.3f, download-sync wait %.3f, sweep issue %.3f, guard moves %d; n %d sweeps %d aeds %d chain passes %ld\n",
            d.st.total_ms * 1e-3, d.st.aed_host_s, d.prof_scan_wait, d.prof_dl_wait, d.prof_issue, d.prof_guard_moves,
            n, d.st.sweeps, d.st.aeds, d.chain_passes);
    if (stats) *stats = d.st;
    return rc;
}

} // namespace sn

#ifdef SN_TEST_HOOKS   // compiled into libstarneig_amd_test.so only (csrc/Makefile), never into the product library
// ---- measurement hook (NOT part of the public C-ABI; scratch/chase_bench.py): average duration
// of one schur_chase_ulds_kernel launch with `chains` full windows on a random Hessenberg matrix
extern "C" __attribute__((visibility("default")))
double sn_internal_chase_bench(int chains, int reps, int dbg)
{
    using namespace sn;
    int const ws_ = WS_MAX, nbc = NB_MAX, adv = ws_ - 1 - 3 * nbc, gap = divceil(ws_ + adv, adv);
    int const n = ws_ + adv * (gap * (chains - 1) + 4) + 200, ld = (int)roundup(n, 16);
    double *H0, *H, *U, *sr, *si;
    SN_HIP_CHECK(hipMalloc((void **)&H0, (size_t)ld * n * 8)); SN_HIP_CHECK(hipMalloc((void **)&H, (size_t)ld * n * 8));
    SN_HIP_CHECK(hipMalloc((void **)&U, (size_t)chains * WS_MAX * WS_MAX * 8));
    SN_HIP_CHECK(hipMalloc((void **)&sr, (size_t)2 * nbc * chains * 8)); SN_HIP_CHECK(hipMalloc((void **)&si, (size_t)2 * nbc * chains * 8));
    lcg_fill(nullptr, n, n, 7u, 1, H0, ld);
    std::vector<double> h((size_t)ld * n);
    SN_HIP_CHECK(hipMemcpy(h.data(), H0, h.size() * 8, hipMemcpyDeviceToHost));
    for (int c = 0; c < n; c++) for (int r = c + 2; r < n; r++) h[(size_t)c * ld + r] = 0.0;
    // bulges in flight: a dense 3x3 bump every 3 columns inside every window position
    for (int c = 0; c + 3 < n; c++) { h[(size_t)c * ld + c + 2] = 0.3; if (c % 3 == 0) h[(size_t)c * ld + c + 3] = 0.2; }
    SN_HIP_CHECK(hipMemcpy(H0, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    std::vector<double> shr(2 * nbc * chains), shi(2 * nbc * chains, 0.0);
    for (size_t k = 0; k < shr.size(); k++) shr[k] = 0.1 + 0.01 * k;
    SN_HIP_CHECK(hipMemcpy(sr, shr.data(), shr.size() * 8, hipMemcpyHostToDevice));
    SN_HIP_CHECK(hipMemcpy(si, shi.data(), shi.size() * 8, hipMemcpyHostToDevice));
    int const size = n, spc = divceil(size - ws_, adv) + 1;
    SweepStep step{0, n, ws_, nbc, adv, gap, nbc * chains, spc, 0, 0, 0};
    step.t = gap * (chains - 1) + 2; step.cmin = 0; step.ntasks = chains;     // every chain mid-flight
    hipEvent_t e0, e1; SN_HIP_CHECK(hipEventCreate(&e0)); SN_HIP_CHECK(hipEventCreate(&e1));
    double total = 0.0;
    for (int r = 0; r < reps + 1; r++) {
        SN_HIP_CHECK(hipMemcpy(H, H0, (size_t)ld * n * 8, hipMemcpyDeviceToDevice));
        SN_HIP_CHECK(hipEventRecord(e0, nullptr));
        auto go = [&](auto kern) {
            SN_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CHASE_LDS_BYTES_WU));
            hipLaunchKernelGGL(kern, dim3(chains), dim3(CHASE_THREADS), CHASE_LDS_BYTES_WU, nullptr, step, H, ld, U, sr, si);
        };
        switch (dbg) {
            case 1: go(schur_chase_dbg_kernel<1>); break;
            case 2: go(schur_chase_dbg_kernel<2>); break;
            case 3: go(schur_chase_dbg_kernel<3>); break;
            case 4: go(schur_chase_dbg_kernel<4>); break;
            case 7: go(schur_chase_dbg_kernel<7>); break;
            default: go(schur_chase_ulds_kernel);
        }
        SN_HIP_CHECK(hipEventRecord(e1, nullptr));
        SN_HIP_CHECK(hipEventSynchronize(e1));
        float ms; SN_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) total += ms;
    }
    SN_HIP_CHECK(hipFree(H0)); SN_HIP_CHECK(hipFree(H)); SN_HIP_CHECK(hipFree(U)); SN_HIP_CHECK(hipFree(sr)); SN_HIP_CHECK(hipFree(si));
    return total / reps * 1e3;      // microseconds
}

// ---- measurement hook (scratch/update_bench.py): average duration of one lazy Q update launch
// (schur_update_kernel<3>: Q(:, window) <- Q(:, window) U for `chains` windows) on an nq-row Q, alone on the GPU
extern "C" __attribute__((visibility("default")))
double sn_internal_qupdate_bench(int nq, int chains, int reps, int rbm)
{
    using namespace sn;
    int const ws_ = WS_MAX, nbc = NB_MAX, adv = ws_ - 1 - 3 * nbc, gap = divceil(ws_ + adv, adv);
    int const n = ws_ + adv * (gap * (chains - 1) + 4) + 200, ld = (int)roundup(nq, 16);
    double *Q, *U;
    SN_HIP_CHECK(hipMalloc((void **)&Q, (size_t)ld * n * 8));
    SN_HIP_CHECK(hipMalloc((void **)&U, (size_t)chains * WS_MAX * WS_MAX * 8));
    lcg_fill(nullptr, nq, n, 7u, 1, Q, ld);
    for (int k = 0; k < chains; k++) set_matrix(nullptr, WS_MAX, WS_MAX, 0.0, 1.0, U + (size_t)k * WS_MAX * WS_MAX, WS_MAX);
    SN_HIP_CHECK(hipFuncSetAttribute((const void *)schur_update_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, UPDATE_LDS_BYTES_R));
    int const spc = divceil(n - ws_, adv) + 1;
    SweepStep step{0, n, ws_, nbc, adv, gap, nbc * chains, spc, 0, 0, 0};
    step.t = gap * (chains - 1) + 2; step.cmin = 0; step.ntasks = chains;
    constexpr int lds64 = GemmCfg<64, WS_MAX, 16, false, false>::LDS_BYTES;
    hipEvent_t e0, e1; SN_HIP_CHECK(hipEventCreate(&e0)); SN_HIP_CHECK(hipEventCreate(&e1));
    double total = 0.0;
    for (int r = 0; r < reps + 1; r++) {
        SN_HIP_CHECK(hipEventRecord(e0, nullptr));
        if (rbm == 64)
            hipLaunchKernelGGL((schur_update_kernel<3, 64>), dim3(divceil(nq, 64), chains), dim3(256),
                lds64, nullptr, step, (double *)nullptr, 0, Q, ld, nq, U, 0, nq);
        else
            hipLaunchKernelGGL(schur_update_kernel<3>, dim3(divceil(nq, 128), chains), dim3(256), UPDATE_LDS_BYTES_R, nullptr,
                step, (double *)nullptr, 0, Q, ld, nq, U, 0, nq);
        SN_HIP_CHECK(hipEventRecord(e1, nullptr));
        SN_HIP_CHECK(hipEventSynchronize(e1));
        float ms; SN_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) total += ms;
    }
    SN_HIP_CHECK(hipFree(Q)); SN_HIP_CHECK(hipFree(U));
    return total / reps * 1e3;
}
#endif  // SN_TEST_HOOKS
