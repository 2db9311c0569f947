// Window bookkeeping shared by the standard (schur.hip) and generalized (schur_gep.hip)
// multi-shift sweeps.
#pragma once
#include "common.h"
#include <cmath>

namespace sn {

struct ChaseTask {
    int lo;         // first row/column of the window in H
    int n;          // window size
    int nb;         // bulges in this chain
    int shift_off;  // index of the chain's first shift
    int flags;      // 1 = introduce, 2 = finalize
    int right;      // column where the trailing bulge of the chain stops (non-finalize windows)
};

// One step of a sweep: every chain in flight advances by one window.  The window of the
// k-th active chain is a pure function of these integers, so the chase kernel and the update
// kernels derive it on the device (no per-step task upload).
struct SweepStep {
    int ilo, ihi;           // active block
    int ws, nbc, adv, gap;  // window size, bulges per chain, columns per step, steps between chains
    int nbulges;            // total bulges of the sweep
    int steps_per_chain;
    int t;                  // step index
    int cmin, ntasks;       // chains cmin .. cmin+ntasks-1 are in flight
};

__host__ __device__ inline ChaseTask make_task(SweepStep const &st, int k)
{
    int const c = st.cmin + k, p = st.t - c * st.gap;
    ChaseTask task;
    task.lo = st.ilo + p * st.adv;
    int const rem = st.nbulges - c * st.nbc;
    task.nb = rem < st.nbc ? rem : st.nbc;
    task.shift_off = 2 * c * st.nbc;
    task.flags = (p == 0) ? 1 : 0;
    if (task.lo + st.ws >= st.ihi) { task.flags |= 2; task.n = st.ihi - task.lo; }
    else task.n = st.ws;
    task.right = st.adv;
    return task;
}

// Householder reflector I - tau [1;v1;v2][1;v1;v2]^T mapping x to beta e1 (len 2 or 3).
// x is scaled by its largest entry first: bulges that pass an (almost) converged part of
// the matrix shrink to 1e-150 and below, where x^2 would lose its bits to underflow
// (LAPACK dlarfg handles the same situation with its safmin rescaling loop).
__host__ __device__ __forceinline__ void small_reflector(int len, double const *x,
    double &beta, double &v1, double &v2, double &tau)
{
    double x0 = x[0], x1 = x[1], x2 = len == 3 ? x[2] : 0.0;
    double const m = fmax(fabs(x0), fmax(fabs(x1), fabs(x2)));
    // (entries below 1e-290 cannot be scaled safely and are dropped: they are far below any
    // deflation threshold)
    if ((x1 == 0.0 && x2 == 0.0) || !(m > 1e-290) || !(fmax(fabs(x1), fabs(x2)) > 1e-290)) {
        beta = x0; v1 = v2 = 0.0; tau = 0.0; return;
    }
    // Scaling by a POWER OF TWO: exact.  (Round 1 multiplied by a rounded 1/m: three independent
    // relative errors in the scaled entries, i.e. a reflector that annihilates a slightly different
    // vector than the one in the matrix.  Over the ~10^4 reflector applications a column of Q sees
    // that was the larger part of the rounding error: n = 20000 random dense 250 u -> 92 u
    // residual, all-ones Hessenberg n = 8000 875 u -> 180 u; measured against the same code with
    // only this line changed.)  tau and the scaling of v follow LAPACK dlarfg; a single shared
    // quotient for both was tried and is LESS accurate (139 u / 264 u on the same two inputs).
    int const e = ilogb(m);
    double a = scalbn(x0, -e), b1 = scalbn(x1, -e), b2 = scalbn(x2, -e);
    double bs = -copysign(sqrt(a * a + b1 * b1 + b2 * b2), a);
    tau = (bs - a) / bs;
    double const sc = 1.0 / (a - bs);
    v1 = b1 * sc; v2 = b2 * sc;
    beta = scalbn(bs, e);
}

} // namespace sn
