// TEST / BENCH INFRASTRUCTURE ONLY (CPU port for bench.py's cpu_baseline leg; never linked into or
// called by the product path).
//
// A multi-threaded host restatement of the reference's Schur leg -- small-bulge multishift QR with
// aggressive early deflation (schur/core.c:2226-2336 state machine; :668-764 chains of tightly packed
// bulges moved window by window; schur/cpu_utils.c:1168-1810 the 3x3-reflector chase inside a window;
// common/cpu.c:54-162 the off-diagonal updates X <- lQ^T X, X <- X lQ as matrix products) -- so that the
// CPU baseline of bench.py is the reference's ALGORITHM on the host cores, not only LAPACK's.
//   * the chase of a chain through a diagonal window and the accumulation of its orthogonal factor U:
//     this file, sequential (it is the critical path of the reference as well);
//   * the updates with U of the columns right of the window, the rows above it and Q: this file,
//     OpenMP over column / row blocks -- the bulk of the flops;
//   * the AED window kernel (Schur form of the window, deflation test with the spike, reordering of the
//     undeflatable blocks, re-reduction to Hessenberg form: cpu_utils.c:2837-3046) and the small-block
//     solver (cpu_utils.c:2426-2516, LAPACK dhseqr there) are CALLBACKS: bench.py passes the host-only
//     window kernels of starneig_amd/csrc/schur_host.hip (test library), the CPU code the product itself
//     runs for these two steps.
// One chain at a time (the reference overlaps several on different workers): parallelism is inside the
// updates.  Deflation criterion: the reference's norm-stable one, |entry| < u ||H||_F.
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

typedef int (*port_aed_fn)(int nw, double *T, int ldt, double *Z, int ldz, double sub, double thres,
    double *spike, double *sr, double *si, int *out3);
typedef int (*port_small_fn)(int n, double *T, int ldt, double *Z, int ldz, double *wr, double *wi);

#define H_(i, j) H[(size_t)(j) * ldh + (i)]
#define Q_(i, j) Q[(size_t)(j) * ldq + (i)]
#define U_(i, j) U[(size_t)(j) * ldu + (i)]

// X(w x ncols) <- U^T X
static void update_left(int w, int ncols, const double *U, int ldu, double *X, int ldx)
{
    if (ncols <= 0) return;
    #pragma omp parallel
    {
        double *buf = malloc((size_t)w * sizeof(double));
        #pragma omp for schedule(static)
        for (int c = 0; c < ncols; c++) {
            double *x = X + (size_t)c * ldx;
            memcpy(buf, x, (size_t)w * sizeof(double));
            for (int i = 0; i < w; i++) {
                const double *u = U + (size_t)i * ldu;
                double s = 0.0;
                #pragma omp simd reduction(+:s)
                for (int k = 0; k < w; k++) s += u[k] * buf[k];
                x[i] = s;
            }
        }
        free(buf);
    }
}

// X(nrows x w) <- X U
static void update_right(int nrows, int w, double *X, int ldx, const double *U, int ldu)
{
    if (nrows <= 0) return;
    int const RB = 64;
    #pragma omp parallel
    {
        double *buf = malloc((size_t)RB * w * sizeof(double));
        #pragma omp for schedule(static)
        for (int r0 = 0; r0 < nrows; r0 += RB) {
            int const nr = nrows - r0 < RB ? nrows - r0 : RB;
            for (int k = 0; k < w; k++) memcpy(buf + (size_t)k * RB, X + (size_t)k * ldx + r0, (size_t)nr * sizeof(double));
            for (int j = 0; j < w; j++) {
                double *out = X + (size_t)j * ldx + r0;
                for (int r = 0; r < nr; r++) out[r] = 0.0;
                for (int k = 0; k < w; k++) {
                    double const ukj = U_(k, j);
                    const double *b = buf + (size_t)k * RB;
                    #pragma omp simd
                    for (int r = 0; r < nr; r++) out[r] += b[r] * ukj;
                }
            }
        }
        free(buf);
    }
}

// LAPACK dlarfg on a vector of length nr <= 3: (I - tau v v^T) x = beta e1, v[0] = 1
static void house_small(int nr, const double *x, double *v, double *tau, double *beta)
{
    double xn = 0.0;
    for (int i = 1; i < nr; i++) xn = hypot(xn, x[i]);
    v[0] = 1.0;
    if (xn == 0.0) { *tau = 0.0; *beta = x[0]; for (int i = 1; i < nr; i++) v[i] = 0.0; return; }
    double const b = -copysign(hypot(x[0], xn), x[0]);
    *tau = (b - x[0]) / b;
    double const sc = 1.0 / (x[0] - b);
    for (int i = 1; i < nr; i++) v[i] = x[i] * sc;
    *beta = b;
}

// One multishift sweep over the active block [ilo, ihi) with the shift pairs (sr, si)[0:ns): chains of
// at most `mmax` bulges, 3 columns apart, moved through diagonal windows of W rows; every window's
// factor U goes to the rest of H and to Q as matrix products.  Returns the flops of those products.
static double sweep(int n, double *H, int ldh, double *Q, int ldq, int ilo, int ihi, int ns,
    const double *sr, const double *si, int W, double *U)
{
    double flops = 0.0;
    int const ldu = W;
    int mmax = (W - 6) / 3;
    if (mmax < 1) mmax = 1;
    int *pos = malloc(sizeof(int) * mmax), *done = malloc(sizeof(int) * mmax);
    for (int b0 = 0; b0 < ns / 2; b0 += mmax) {
        int const m = (ns / 2 - b0 < mmax) ? ns / 2 - b0 : mmax;
        for (int k = 0; k < m; k++) { pos[k] = ilo - 1; done[k] = 0; }
        int w0 = ilo;
        for (;;) {
            int alive = 0;
            for (int k = 0; k < m; k++) alive += !done[k];
            if (!alive) break;
            int const w1 = (w0 + W < ihi) ? w0 + W : ihi, ww = w1 - w0;
            for (int j = 0; j < ww; j++) for (int i = 0; i < ww; i++) U_(i, j) = (i == j) ? 1.0 : 0.0;
            int progress = 1, moved = 0;
            while (progress) {
                progress = 0;
                for (int k = 0; k < m; k++) {
                    if (done[k]) continue;
                    int const j = pos[k];
                    if (k > 0 && !done[k - 1] && pos[k - 1] < j + 4) continue;      // the bulge ahead is too close
                    if (j == ilo - 1) { if (w0 != ilo) continue; }
                    else if (j < w0) continue;
                    int const nr = (ihi - j - 1 < 3) ? ihi - j - 1 : 3;
                    if (nr < 2) { done[k] = 1; progress = 1; continue; }
                    int const rmax = (j + nr + 1 < ihi - 1) ? j + nr + 1 : ihi - 1;
                    if (rmax > w1 - 1) continue;                                    // the fill would leave the window
                    double x[3], v[3], tau, beta;
                    if (j == ilo - 1) {
                        // first column of (H - s1)(H - s2) (LAPACK dlaqr1; cpu_utils.c:880-918)
                        double const s = sr[2 * (b0 + k)] + sr[2 * (b0 + k) + 1];
                        double const p = sr[2 * (b0 + k)] * sr[2 * (b0 + k) + 1] - si[2 * (b0 + k)] * si[2 * (b0 + k) + 1];
                        double const h11 = H_(ilo, ilo), h21 = H_(ilo + 1, ilo), h12 = H_(ilo, ilo + 1), h22 = H_(ilo + 1, ilo + 1);
                        x[0] = h11 * h11 + h12 * h21 - s * h11 + p;
                        x[1] = h21 * (h11 + h22 - s);
                        x[2] = (nr == 3) ? h21 * H_(ilo + 2, ilo + 1) : 0.0;
                    } else
                        for (int i = 0; i < nr; i++) x[i] = H_(j + 1 + i, j);
                    house_small(nr, x, v, &tau, &beta);
                    if (j >= ilo) { H_(j + 1, j) = beta; for (int i = 1; i < nr; i++) H_(j + 1 + i, j) = 0.0; }
                    if (tau != 0.0) {
                        // left: rows j+1 .. j+nr, columns j+1 .. w1-1
                        for (int c = j + 1; c < w1; c++) {
                            double *col = &H_(j + 1, c);
                            double t = 0.0;
                            for (int i = 0; i < nr; i++) t += v[i] * col[i];
                            t *= tau;
                            for (int i = 0; i < nr; i++) col[i] -= t * v[i];
                        }
                        // right: columns j+1 .. j+nr, rows w0 .. rmax; and the window's factor U
                        for (int r = w0; r <= rmax; r++) {
                            double t = 0.0;
                            for (int i = 0; i < nr; i++) t += H_(r, j + 1 + i) * v[i];
                            t *= tau;
                            for (int i = 0; i < nr; i++) H_(r, j + 1 + i) -= t * v[i];
                        }
                        for (int r = 0; r < ww; r++) {
                            double t = 0.0;
                            for (int i = 0; i < nr; i++) t += U_(r, j + 1 - w0 + i) * v[i];
                            t *= tau;
                            for (int i = 0; i < nr; i++) U_(r, j + 1 - w0 + i) -= t * v[i];
                        }
                    }
                    pos[k] = j + 1;
                    if (pos[k] >= ihi - 2) done[k] = 1;
                    progress = 1; moved = 1;
                }
            }
            // the rest of H and Q see the window through U
            update_left(ww, n - w1, U, ldu, &H_(w0, w1), ldh);
            update_right(w0, ww, &H_(0, w0), ldh, U, ldu);
            update_right(n, ww, &Q_(0, w0), ldq, U, ldu);
            flops += 2.0 * ww * (double)ww * ((n - w1) + w0 + n);
            int next = ihi;
            for (int k = 0; k < m; k++) if (!done[k] && pos[k] < next) next = pos[k];
            if (next <= w0 && !moved) break;            // (cannot happen for m <= (W - 6) / 3: no endless loop on a logic error)
            w0 = (next > ilo) ? next : ilo;
            if (w0 > ihi - 2) break;
        }
    }
    free(pos); free(done);
    return flops;
}

// H (upper Hessenberg, n x n) -> real Schur form, Q <- Q U.  nw / ns: AED window and shift count; W: chase
// window.  stats[0..3] = sweeps, AEDs, update flops, seconds inside the AED callback.  Returns 0, or 1 if the
// iteration limit was hit.
int oracle_msqr_port(int n, double *H, int ldh, double *Q, int ldq, int nw, int ns, int W, int small_limit,
    port_aed_fn aed, port_small_fn small, double *stats)
{
    double hn = 0.0;
    for (int j = 0; j < n; j++) { int const top = (j + 2 < n) ? j + 2 : n; for (int i = 0; i < top; i++) hn = hypot(hn, H_(i, j)); }
    double const thres = DBL_EPSILON * hn;
    int const wmax = (nw > small_limit ? nw : small_limit) + 8;
    double *T = malloc(sizeof(double) * wmax * wmax), *Z = malloc(sizeof(double) * wmax * wmax);
    double *U = malloc(sizeof(double) * W * W);
    double *spike = malloc(sizeof(double) * wmax), *sr = malloc(sizeof(double) * 2 * wmax), *si = malloc(sizeof(double) * 2 * wmax);
    double *wr = malloc(sizeof(double) * wmax), *wi = malloc(sizeof(double) * wmax);
    int ihi = n, iter = 0, rc = 0, stagnation = 0;
    double sweeps = 0, aeds = 0, flops = 0, t_aed = 0;
    while (ihi > 0) {
        int ilo = ihi - 1;
        while (ilo > 0) {
            if (fabs(H_(ilo, ilo - 1)) < thres) { H_(ilo, ilo - 1) = 0.0; break; }
            ilo--;
        }
        int const size = ihi - ilo;
        if (size == 1) { ihi = ilo; continue; }
        int w = 0, lo = 0, nd = 0;
        int out3[3] = {0, 0, 0};
        if (size <= small_limit) {
            w = size; lo = ilo;
            for (int j = 0; j < w; j++) for (int i = 0; i < w; i++) { T[(size_t)j * wmax + i] = (i <= j + 1) ? H_(lo + i, lo + j) : 0.0; Z[(size_t)j * wmax + i] = (i == j); }
            if (small(w, T, wmax, Z, wmax, wr, wi) != 0) { rc = 1; break; }
            nd = w;
        } else {
            if (++iter > 30 * n) { rc = 1; break; }
            w = (nw < size) ? nw : size; lo = ihi - w;
            double const sub = (lo > ilo) ? H_(lo, lo - 1) : 0.0;
            for (int j = 0; j < w; j++) for (int i = 0; i < w; i++) T[(size_t)j * wmax + i] = (i <= j + 1) ? H_(lo + i, lo + j) : 0.0;
            double const t0 = omp_get_wtime();
            aed(w, T, wmax, Z, wmax, sub, thres, spike, sr, si, out3);
            t_aed += omp_get_wtime() - t0; aeds++;
            nd = out3[0];
            if (nd > 0 && lo > ilo) H_(lo, lo - 1) = spike[0];
        }
        if (nd > 0) {
            for (int j = 0; j < w; j++) for (int i = 0; i < w; i++) H_(lo + i, lo + j) = T[(size_t)j * wmax + i];
            update_left(w, n - (lo + w), Z, wmax, &H_(lo, lo + w), ldh);
            update_right(lo, w, &H_(0, lo), ldh, Z, wmax);
            update_right(n, w, &Q_(0, lo), ldq, Z, wmax);
            flops += 2.0 * w * (double)w * ((n - lo - w) + lo + n);
            ihi -= nd; stagnation = 0;
        } else stagnation++;
        if (size <= small_limit) continue;
        if (ihi - ilo <= small_limit) continue;
        if (100 * nd > 40 * w) continue;                      // nibble rule (process_args.c:356)
        int nsh = (out3[1] < ns) ? out3[1] : ns;
        nsh -= nsh % 2;
        if (nsh < 2 || (stagnation > 0 && stagnation % 6 == 0)) {
            // exceptional shifts (LAPACK dlaqr0): from the sub-diagonal magnitudes at the bottom of the block
            nsh = 2;
            double const ss = fabs(H_(ihi - 1, ihi - 2)) + (ihi - 3 >= ilo ? fabs(H_(ihi - 2, ihi - 3)) : 0.0);
            sr[0] = sr[1] = 0.75 * ss + H_(ihi - 1, ihi - 1); si[0] = 0.66 * ss; si[1] = -si[0];
        }
        if (stagnation > 60) { rc = 1; break; }
        flops += sweep(n, H, ldh, Q, ldq, ilo, ihi, nsh, sr, si, W, U);
        sweeps++;
    }
    free(T); free(Z); free(U); free(spike); free(sr); free(si); free(wr); free(wi);
    if (stats) { stats[0] = sweeps; stats[1] = aeds; stats[2] = flops; stats[3] = t_aed; }
    return rc;
}

// Thread count of the OpenMP regions of the oracle (bench.py's cpu_baseline: the environment variable is read
// when the OpenMP runtime starts, which in a process that imported torch was long ago).  Returns the count in force.
int oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n; return 1;
#endif
}
