/*
 * CPU ORACLE of the Hessenberg-triangular reduction -- TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load
 * or call this file; the product (starneig_amd/) never does.
 *
 * What it restates: starneig_GEP_SM_HessenbergTriangular (reference src/wrappers/lapack.c:45-176),
 * which is a sequence of LAPACK calls:
 *     dgeqrf(B)                 B = Q0 R                     (lapack.c:143)
 *     dormqr('L','T')           A <- Q0^T A                  (lapack.c:148)
 *     dormqr('R','N')           Q <- Q Q0                    (lapack.c:153)
 *     B <- R  (strictly lower part cleared)                  (lapack.c:158-160)
 *     dgghd3('V','V')           (A, R) -> (H, T), Q, Z       (lapack.c:163)
 * LAPACK is a third-party dependency that is not vendored under /root/reference (the CMake
 * build links the system's OpenBLAS/LAPACK), so the oracle restates the published algorithms
 * in plain loops: Householder QR with the dlarfg conventions (unblocked dgeqr2 / dorm2r, which
 * dgeqrf / dormqr reproduce up to rounding), and the rotation-based reduction of Moler & Stewart
 * in the order and with the dlartg (LAPACK >= 3.10) conventions of dgghrd.  dgghd3 is the blocked
 * form of the same reduction; the decomposition is not unique across the two (different rotation
 * groupings), so parity with dgghd3 is stated on the backward error and the structure, while
 * parity with dgghrd is elementwise.
 *
 * Pinning: tests/golden/ht_lcg2019_n*.npz hold what LAPACK (scipy's bundled OpenBLAS: dgeqrf,
 * dormqr, dgghrd, reached through ctypes by tests/golden/make_golden_ht.py) computes for the
 * reference test driver's input (two generate_random_fullpos matrices on one LCG stream,
 * test/hessenberg/experiment.c:102-106); tests/test_oracle_ht.py checks this file against them.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EL(M,ld,i,j) ((M)[(size_t)(j)*(ld)+(i)])

/* LAPACK dlarfg: H = I - tau v v^T, v[0] = 1, H (alpha, x)^T = (beta, 0)^T */
static double larfg(int n, double *alpha, double *x, int incx)
{
    double s = 0.0;
    for (int i = 0; i < n-1; i++) s += x[(size_t)i*incx]*x[(size_t)i*incx];
    if (n <= 1 || s == 0.0) return 0.0;
    double const beta = -copysign(sqrt(*alpha * *alpha + s), *alpha);
    double const tau = (beta - *alpha) / beta;
    double const sc = 1.0 / (*alpha - beta);
    for (int i = 0; i < n-1; i++) x[(size_t)i*incx] *= sc;
    *alpha = beta;
    return tau;
}

/* LAPACK 3.10+ dlartg (la_lartg.f90), unscaled branch: [c s; -s c] (f, g)^T = (r, 0)^T */
void oracle_lartg(double f, double g, double *c, double *s, double *r)
{
    if (g == 0.0) { *c = 1.0; *s = 0.0; *r = f; }
    else if (f == 0.0) { *c = 0.0; *s = copysign(1.0, g); *r = fabs(g); }
    else {
        double const d = sqrt(f*f + g*g);
        *c = fabs(f) / d;
        *r = copysign(d, f);
        *s = g / *r;
    }
}

/* x' = c x + s y, y' = c y - s x (BLAS drot) */
static void rot(int n, double *x, int incx, double *y, int incy, double c, double s)
{
    for (int i = 0; i < n; i++) {
        double const a = x[(size_t)i*incx], b = y[(size_t)i*incy];
        x[(size_t)i*incx] = c*a + s*b;
        y[(size_t)i*incy] = c*b - s*a;
    }
}

/* lapack.c:143-160: B = Q0 R, A <- Q0^T A, Q <- Q Q0, B <- R */
int oracle_ht_qr(int n, double *A, int ldA, double *B, int ldB, double *Q, int ldQ)
{
    double *v = malloc((size_t)n*sizeof(double));
    if (!v) return -1;
    for (int j = 0; j < n; j++) {
        int const m = n - j;
        double const tau = larfg(m, &EL(B,ldB,j,j), &EL(B,ldB,j+1 < n ? j+1 : j,j), 1);
        v[0] = 1.0;
        for (int i = 1; i < m; i++) { v[i] = EL(B,ldB,j+i,j); EL(B,ldB,j+i,j) = 0.0; }
        if (tau == 0.0) continue;
        /* H_j from the left on B(j:n, j+1:n) and A(j:n, :) */
        for (int c = j+1; c < n; c++) {
            double w = 0.0;
            for (int i = 0; i < m; i++) w += v[i]*EL(B,ldB,j+i,c);
            w *= tau;
            for (int i = 0; i < m; i++) EL(B,ldB,j+i,c) -= v[i]*w;
        }
        for (int c = 0; c < n; c++) {
            double w = 0.0;
            for (int i = 0; i < m; i++) w += v[i]*EL(A,ldA,j+i,c);
            w *= tau;
            for (int i = 0; i < m; i++) EL(A,ldA,j+i,c) -= v[i]*w;
        }
        /* H_j from the right on Q(:, j:n) */
        for (int r = 0; r < n; r++) {
            double w = 0.0;
            for (int i = 0; i < m; i++) w += EL(Q,ldQ,r,j+i)*v[i];
            w *= tau;
            for (int i = 0; i < m; i++) EL(Q,ldQ,r,j+i) -= w*v[i];
        }
    }
    free(v);
    return 0;
}

/* dgghrd('V','V', ilo = 1, ihi = n): B upper triangular on entry */
int oracle_ht_reduce(int n, double *A, int ldA, double *B, int ldB,
    double *Q, int ldQ, double *Z, int ldZ)
{
    for (int jc = 0; jc + 2 < n; jc++) {
        for (int jr = n-1; jr >= jc+2; jr--) {
            double c, s, r;
            /* rows jr-1, jr: annihilate A(jr, jc) */
            oracle_lartg(EL(A,ldA,jr-1,jc), EL(A,ldA,jr,jc), &c, &s, &r);
            EL(A,ldA,jr-1,jc) = r; EL(A,ldA,jr,jc) = 0.0;
            rot(n-jc-1, &EL(A,ldA,jr-1,jc+1), ldA, &EL(A,ldA,jr,jc+1), ldA, c, s);
            rot(n-jr+1, &EL(B,ldB,jr-1,jr-1), ldB, &EL(B,ldB,jr,jr-1), ldB, c, s);
            rot(n, &EL(Q,ldQ,0,jr-1), 1, &EL(Q,ldQ,0,jr), 1, c, s);
            /* columns jr, jr-1: annihilate B(jr, jr-1) */
            oracle_lartg(EL(B,ldB,jr,jr), EL(B,ldB,jr,jr-1), &c, &s, &r);
            EL(B,ldB,jr,jr) = r; EL(B,ldB,jr,jr-1) = 0.0;
            rot(n, &EL(A,ldA,0,jr), 1, &EL(A,ldA,0,jr-1), 1, c, s);
            rot(jr, &EL(B,ldB,0,jr), 1, &EL(B,ldB,0,jr-1), 1, c, s);
            rot(n, &EL(Z,ldZ,0,jr), 1, &EL(Z,ldZ,0,jr-1), 1, c, s);
        }
    }
    return 0;
}

int oracle_hessenberg_triangular(int n, double *A, int ldA, double *B, int ldB,
    double *Q, int ldQ, double *Z, int ldZ)
{
    int rc = oracle_ht_qr(n, A, ldA, B, ldB, Q, ldQ);
    if (rc) return rc;
    return oracle_ht_reduce(n, A, ldA, B, ldB, Q, ldQ, Z, ldZ);
}
