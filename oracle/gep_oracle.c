/*
 * CPU ORACLE of the generalized Schur (QZ) leg -- TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load
 * or call this file; the product (starneig_amd/) never does.
 *
 * What it restates: the reduction of a Hessenberg-triangular pencil (H, R) to generalized
 * real Schur form (S, T) with accumulated Q, Z that starneig_GEP_SM_Schur performs
 * (reference src/schur/interface.c:241-300 -> core.c:2342).  On a CPU the reference reduces
 * every window / small pencil with LAPACK dhgeqz (schur/cpu_utils.c:2248-2309, pencils of
 * at most 64 rows) or with its own sequential multi-shift QZ loop (:3185-3371) and
 * standardises 2x2 blocks with dlagv2 (:801-850); both are the blocked/multi-shift forms
 * of the QZ iteration of Moler & Stewart.  LAPACK (OpenBLAS 0.3.x as linked by the
 * reference's CMake build) is a third-party dependency that is not vendored under
 * /root/reference, so the oracle restates the published algorithm it implements: the
 * implicit double-shift QZ step (Golub & Van Loan, Matrix Computations, Alg. 7.7.2/7.7.3)
 * with LAPACK's deflation test |h(k,k-1)| <= ulp (|h(k-1,k-1)| + |h(k,k)|) and the
 * dlagv2 conventions for 2x2 blocks (real pair -> upper triangular; complex pair -> T
 * diagonal, positive, t11 >= t22).
 *
 * Pinning: tests/golden/gep_lcg2019_n*.npz hold the eigenvalues LAPACK dhgeqz (through
 * scipy.linalg.qz, i.e. the routine the reference itself calls) computes for the
 * reference test driver's random pencils (test/schur/experiment.c:203-207, LCG seed 2019);
 * tests/test_oracle_gep.py checks this oracle against them and against its own invariants.
 *
 * The QZ iteration is not unique (different shift strategies give different, equally
 * valid Schur forms), so parity is stated on what IS unique: the generalized eigenvalues
 * and the backward error / structure of the decomposition.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define MIN(a,b) ((a) < (b) ? (a) : (b))
#define MAX(a,b) ((a) > (b) ? (a) : (b))
#define EL(M,ld,i,j) ((M)[(size_t)(j)*(ld)+(i)])

int oracle_prand(void);
#define PRAND_MAX 0x7fffffff

/* test/common/init.c:122-138 */
void oracle_fill_random_uptriag(int n, double *A, int ld)
{
    for (int j = 0; j < n; j++) {
        int end = MIN(n, j+1);
        for (int i = 0; i < end; i++)
            A[(size_t)j*ld+i] = 2.0*(1.0*oracle_prand()/PRAND_MAX)-1.0;
        for (int i = end; i < n; i++)
            A[(size_t)j*ld+i] = 0.0;
    }
}

/* Householder vector for x (len 2 or 3): (I - tau v v^T) x = beta e1, v[0] = 1 */
static void house(int len, double const *x, double *v, double *tau)
{
    double s = 0.0;
    for (int i = 1; i < len; i++) s += x[i]*x[i];
    v[0] = 1.0; v[1] = v[2] = 0.0;
    if (s == 0.0) { *tau = 0.0; return; }
    double mu = sqrt(x[0]*x[0] + s);
    double beta = (x[0] <= 0.0) ? -mu : mu;    /* beta = sign(x0) mu, v0 = x0 + beta */
    double v0 = x[0] + beta;
    for (int i = 1; i < len; i++) v[i] = x[i]/v0;
    *tau = v0/beta;
}

/* rows r..r+len-1, columns c0..c1-1:  M <- (I - tau v v^T) M */
static void apply_left(int len, double const *v, double tau, double *M, int ld, int r, int c0, int c1)
{
    if (tau == 0.0) return;
    for (int c = c0; c < c1; c++) {
        double s = 0.0;
        for (int i = 0; i < len; i++) s += v[i]*EL(M,ld,r+i,c);
        s *= tau;
        for (int i = 0; i < len; i++) EL(M,ld,r+i,c) -= s*v[i];
    }
}

/* columns c..c+len-1, rows r0..r1-1:  M <- M (I - tau v v^T) */
static void apply_right(int len, double const *v, double tau, double *M, int ld, int c, int r0, int r1)
{
    if (tau == 0.0) return;
    for (int r = r0; r < r1; r++) {
        double s = 0.0;
        for (int i = 0; i < len; i++) s += v[i]*EL(M,ld,r,c+i);
        s *= tau;
        for (int i = 0; i < len; i++) EL(M,ld,r,c+i) -= s*v[i];
    }
}

/* Householder P with  x^T P = (0, .., 0, beta):  reflect onto the LAST coordinate
 * (Golub & Van Loan 7.7.2's Z_k1 / Z_k2: zeros are made from the right, to the left of the
 * diagonal entry) */
static void house_last(int len, double const *x, double *v, double *tau)
{
    double y[3], w[3];
    for (int i = 0; i < len; i++) y[i] = x[len-1-i];
    house(len, y, w, tau);
    for (int i = 0; i < len; i++) v[i] = w[len-1-i];
}

static void rot(double f, double g, double *c, double *s)
{
    if (g == 0.0) { *c = 1.0; *s = 0.0; return; }
    double r = hypot(f, g);
    *c = f/r; *s = g/r;
}

/* rows i, k of M over columns c0..c1-1: [x; y] <- [c s; -s c] [x; y] */
static void rot_rows(double *M, int ld, int i, int k, int c0, int c1, double c, double s)
{
    for (int j = c0; j < c1; j++) {
        double x = EL(M,ld,i,j), y = EL(M,ld,k,j);
        EL(M,ld,i,j) = c*x + s*y; EL(M,ld,k,j) = c*y - s*x;
    }
}
/* columns i, k of M over rows r0..r1-1: [x y] <- [x y] [c -s; s c] */
static void rot_cols(double *M, int ld, int i, int k, int r0, int r1, double c, double s)
{
    for (int r = r0; r < r1; r++) {
        double x = EL(M,ld,r,i), y = EL(M,ld,r,k);
        EL(M,ld,r,i) = c*x + s*y; EL(M,ld,r,k) = c*y - s*x;
    }
}

/* Standardise the 2x2 diagonal block at k (dlagv2 conventions).  Returns 1 if the block
 * had real eigenvalues and was split (A(k+1,k) = 0 afterwards), 0 for a complex pair. */
static int standardise(int n, int k, double *A, int lda, double *B, int ldb,
    double *Q, int ldq, double *Z, int ldz)
{
    /* real pair: rotate an eigenvector of the block to e1.  For (nearly) coinciding
     * eigenvalues one pass leaves A(k+1,k) at the sqrt(ulp) level; the passes are orthogonal
     * transformations, so they are repeated until the entry meets the deflation criterion
     * (what dhgeqz achieves by continuing its single-shift iteration on the block) */
    double prev = HUGE_VAL;
    for (int pass = 0; pass < 60; pass++) {
        double a11 = EL(A,lda,k,k), a12 = EL(A,lda,k,k+1), a21 = EL(A,lda,k+1,k), a22 = EL(A,lda,k+1,k+1);
        double b11 = EL(B,ldb,k,k), b12 = EL(B,ldb,k,k+1), b22 = EL(B,ldb,k+1,k+1);
        if (fabs(a21) <= DBL_EPSILON*(fabs(a11) + fabs(a22))) { EL(A,lda,k+1,k) = 0.0; return 1; }
        if (fabs(a21) >= 0.25*prev) {
            /* stagnation at the rounding floor of the block: a few ulp of its largest entry */
            if (fabs(a21) <= 1000.0*DBL_EPSILON*(fabs(a11) + fabs(a12) + fabs(a22))) {
                EL(A,lda,k+1,k) = 0.0; return 1;
            }
            if (pass > 8) break;
        }
        prev = fabs(a21);
        /* det(A - l B) = p l^2 - q l + r */
        double p = b11*b22, q = a11*b22 + a22*b11 - a21*b12, r = a11*a22 - a12*a21;
        double disc = q*q - 4.0*p*r;
        if (!(disc >= 0.0)) break;
        double c, s, zx, zy;
        if (p != 0.0) {
            double sq = sqrt(disc);
            /* the root of smaller magnitude, without cancellation: 2r / (q + sign(q) sqrt(disc)) */
            double den = (q >= 0.0) ? (q + sq) : (q - sq);
            double l = (den != 0.0) ? 2.0*r/den : 0.0;
            double c11 = a11 - l*b11, c12 = a12 - l*b12, c21 = a21, c22 = a22 - l*b22;
            /* null vector from the row of larger norm */
            if (fabs(c11) + fabs(c12) >= fabs(c21) + fabs(c22)) { zx = c12; zy = -c11; }
            else { zx = c22; zy = -c21; }
        } else {
            /* B singular: infinite eigenvalue, z = null vector of B */
            if (b11 == 0.0) { zx = 1.0; zy = 0.0; } else { zx = b12; zy = -b11; }
        }
        /* Z: columns (k,k+1) <- . [c -s; s c] with first column = z/|z| */
        double nz = hypot(zx, zy);
        if (nz == 0.0) { zx = 1.0; zy = 0.0; nz = 1.0; }
        c = zx/nz; s = zy/nz;
        rot_cols(A, lda, k, k+1, 0, k+2, c, s);
        rot_cols(B, ldb, k, k+1, 0, k+2, c, s);
        rot_cols(Z, ldz, k, k+1, 0, n, c, s);
        /* left rotation restoring B's triangular form; A(k+1,k) follows (A z = l B z) */
        double fa = EL(A,lda,k,k), ga = EL(A,lda,k+1,k), fb = EL(B,ldb,k,k), gb = EL(B,ldb,k+1,k);
        if (hypot(fb, gb) > 0.0) rot(fb, gb, &c, &s);
        else rot(fa, ga, &c, &s);
        rot_rows(A, lda, k, k+1, k, n, c, s);
        rot_rows(B, ldb, k, k+1, k, n, c, s);
        rot_cols(Q, ldq, k, k+1, 0, n, c, s);
        EL(B,ldb,k+1,k) = 0.0;
    }
    double b11 = EL(B,ldb,k,k), b12 = EL(B,ldb,k,k+1), b22 = EL(B,ldb,k+1,k+1);
    /* complex pair: B <- U^T B V = diag(s1 >= s2 > 0) by a 2x2 SVD.
     * Step 1: left rotation making B symmetric; step 2: Jacobi rotation diagonalising it. */
    double c1, s1;
    {   /* [c s; -s c] B symmetric: c*b12 + s*b22 = -s*b11  ->  tan = -b12/(b11+b22) */
        double den = b11 + b22, num = -b12;
        if (den == 0.0 && num == 0.0) { c1 = 1.0; s1 = 0.0; }
        else { double h = hypot(den, num); c1 = den/h; s1 = num/h; }
    }
    rot_rows(A, lda, k, k+1, k, n, c1, s1);
    rot_rows(B, ldb, k, k+1, k, n, c1, s1);
    rot_cols(Q, ldq, k, k+1, 0, n, c1, s1);
    {
        double m11 = EL(B,ldb,k,k), m12 = 0.5*(EL(B,ldb,k,k+1) + EL(B,ldb,k+1,k)), m22 = EL(B,ldb,k+1,k+1);
        double c2 = 1.0, s2 = 0.0;
        if (m12 != 0.0) {
            double th = (m22 - m11)/(2.0*m12);
            double t = (th >= 0.0 ? 1.0 : -1.0)/(fabs(th) + sqrt(1.0 + th*th));
            c2 = 1.0/sqrt(1.0 + t*t); s2 = t*c2;
        }
        /* J = [c2 s2; -s2 c2]: B <- J^T B J */
        rot_rows(A, lda, k, k+1, k, n, c2, -s2);
        rot_rows(B, ldb, k, k+1, k, n, c2, -s2);
        rot_cols(Q, ldq, k, k+1, 0, n, c2, -s2);
        rot_cols(A, lda, k, k+1, 0, k+2, c2, -s2);
        rot_cols(B, ldb, k, k+1, 0, k+2, c2, -s2);
        rot_cols(Z, ldz, k, k+1, 0, n, c2, -s2);
    }
    EL(B,ldb,k,k+1) = 0.0; EL(B,ldb,k+1,k) = 0.0;
    /* positive diagonal: flip columns of (A,B,Z) */
    for (int j = k; j <= k+1; j++)
        if (EL(B,ldb,j,j) < 0.0) {
            for (int i = 0; i <= k+1; i++) { EL(A,lda,i,j) = -EL(A,lda,i,j); EL(B,ldb,i,j) = -EL(B,ldb,i,j); }
            for (int i = 0; i < n; i++) EL(Z,ldz,i,j) = -EL(Z,ldz,i,j);
        }
    /* t11 >= t22: swap rows and columns k <-> k+1 (a permutation on both sides) */
    if (EL(B,ldb,k,k) < EL(B,ldb,k+1,k+1)) {
        rot_rows(A, lda, k, k+1, k, n, 0.0, 1.0);
        rot_rows(B, ldb, k, k+1, k, n, 0.0, 1.0);
        rot_cols(Q, ldq, k, k+1, 0, n, 0.0, 1.0);
        rot_cols(A, lda, k, k+1, 0, k+2, 0.0, 1.0);
        rot_cols(B, ldb, k, k+1, 0, k+2, 0.0, 1.0);
        rot_cols(Z, ldz, k, k+1, 0, n, 0.0, 1.0);
        EL(B,ldb,k,k+1) = 0.0; EL(B,ldb,k+1,k) = 0.0;
    }
    return 0;
}

/* (alpha_r + i alpha_i)/beta per diagonal position of a generalized Schur form
 * (the contract of starneig_extract_eigenvalues, schur/cpu_utils.c:3560-3594, GEP branch) */
/* Eigenvalues of a 2 x 2 pencil (A, B), B upper triangular, as (wr1 + i wi) / scale1 and
 * (wr2 - i wi) / scale2: LAPACK DLAG2 restated from its published algorithm (the reference calls it for
 * every 2 x 2 block of a generalized Schur form: common/math.c:148-176, reached from the solver's
 * eigenvalue extraction common/tasks.c:1134, schur/cpu_utils.c:3498 AND from the test hooks'
 * test/common/checks.c:82 -- the same routine on the same block, which is why its `eigenvalues` hook
 * holds to a few u).  C. Van Loan's method: the larger eigenvalue of A B^-1 shifted by the smaller of
 * a11/b11, a22/b22, the smaller one from the determinant; scalings against over / underflow.
 * Compiled without floating-point contraction: the product's host kernel (csrc/schur_host_gep.hip
 * pencil2_dlag2) runs the same operations in the same order, and the two must agree to the last bit
 * on a block whose discriminant nearly vanishes (a rounding there moves the pair by sqrt(u)).
 * Pinned on LAPACK itself: tests/golden/dlag2_cases.npz (tests/golden/make_golden_dlag2.py). */
__attribute__((optimize("fp-contract=off")))
void oracle_dlag2(double const *A, int lda, double const *B, int ldb, double safmin,
    double *scale1, double *scale2, double *wr1, double *wr2, double *wi)
{
    double const fuzzy1 = 1.0 + 1.0e-5;
    double const rtmin = sqrt(safmin), rtmax = 1.0/rtmin, safmax = 1.0/safmin;
    /* scale A */
    double const anorm = MAX(MAX(fabs(A[0]) + fabs(A[1]), fabs(A[lda]) + fabs(A[lda+1])), safmin);
    double const ascale = 1.0/anorm;
    double const a11 = ascale*A[0], a21 = ascale*A[1], a12 = ascale*A[lda], a22 = ascale*A[lda+1];
    /* perturb B if necessary to insure non-singularity */
    double b11 = B[0], b12 = B[ldb], b22 = B[ldb+1];
    double const bmin = rtmin*MAX(MAX(fabs(b11), fabs(b12)), MAX(fabs(b22), rtmin));
    if (fabs(b11) < bmin) b11 = copysign(bmin, b11);
    if (fabs(b22) < bmin) b22 = copysign(bmin, b22);
    /* scale B */
    double const bnorm = MAX(MAX(fabs(b11), fabs(b12) + fabs(b22)), safmin);
    double const bsize = MAX(fabs(b11), fabs(b22));
    double const bscale = 1.0/bsize;
    b11 *= bscale; b12 *= bscale; b22 *= bscale;
    /* larger eigenvalue (AS = A - shift B) */
    double const binv11 = 1.0/b11, binv22 = 1.0/b22;
    double const s1 = a11*binv11, s2 = a22*binv22;
    double as12, abi22, pp, shift, ss;
    if (fabs(s1) <= fabs(s2)) {
        as12 = a12 - s1*b12;
        double const as22 = a22 - s1*b22;
        ss = a21*(binv11*binv22);
        abi22 = as22*binv22 - ss*b12;
        pp = 0.5*abi22;
        shift = s1;
    } else {
        as12 = a12 - s2*b12;
        double const as11 = a11 - s2*b11;
        ss = a21*(binv11*binv22);
        abi22 = -ss*b12;
        pp = 0.5*(as11*binv11 + abi22);
        shift = s2;
    }
    double const qq = ss*as12;
    double discr, r;
    if (fabs(pp*rtmin) >= 1.0) {
        double const t = rtmin*pp;
        discr = t*t + qq*safmin;
        r = sqrt(fabs(discr))*rtmax;
    } else if (pp*pp + fabs(qq) <= safmin) {
        double const t = rtmax*pp;
        discr = t*t + qq*safmax;
        r = sqrt(fabs(discr))*rtmin;
    } else {
        discr = pp*pp + qq;
        r = sqrt(fabs(discr));
    }
    if (discr >= 0.0 || r == 0.0) {
        double const sr = copysign(r, pp);
        double const sum = pp + sr, diff = pp - sr;
        double const wbig = shift + sum;
        double wsmall = shift + diff;
        if (0.5*fabs(wbig) > MAX(fabs(wsmall), safmin)) {
            double const wdet = (a11*a22 - a12*a21)*(binv11*binv22);
            wsmall = wdet/wbig;
        }
        /* the (real) eigenvalue closest to the (2,2) element of A B^-1 goes to wr1 */
        if (pp > abi22) { *wr1 = MIN(wbig, wsmall); *wr2 = MAX(wbig, wsmall); }
        else { *wr1 = MAX(wbig, wsmall); *wr2 = MIN(wbig, wsmall); }
        *wi = 0.0;
    } else {
        *wr1 = shift + pp; *wr2 = *wr1; *wi = r;
    }
    /* further scaling against under / overflow of scale1 and of w B */
    double const c1 = bsize*(safmin*MAX(1.0, ascale));
    double const c2 = safmin*MAX(1.0, bnorm);
    double const c3 = bsize*safmin;
    double const c4 = (ascale <= 1.0 && bsize <= 1.0) ? MIN(1.0, (ascale/safmin)*bsize) : 1.0;
    double const c5 = (ascale <= 1.0 || bsize <= 1.0) ? MIN(1.0, ascale*bsize) : 1.0;
    /* first eigenvalue */
    double wabs = fabs(*wr1) + fabs(*wi);
    double wsize = MAX(MAX(safmin, c1), MAX(fuzzy1*(wabs*c2 + c3), MIN(c4, 0.5*MAX(wabs, c5))));
    if (wsize != 1.0) {
        double const wscale = 1.0/wsize;
        if (wsize > 1.0) *scale1 = (MAX(ascale, bsize)*wscale)*MIN(ascale, bsize);
        else *scale1 = (MIN(ascale, bsize)*wscale)*MAX(ascale, bsize);
        *wr1 *= wscale;
        if (*wi != 0.0) { *wi *= wscale; *wr2 = *wr1; *scale2 = *scale1; }
    } else {
        *scale1 = ascale*bsize;
        *scale2 = *scale1;
    }
    /* second eigenvalue, if real */
    if (*wi == 0.0) {
        wabs = fabs(*wr2);
        wsize = MAX(MAX(safmin, c1), MAX(fuzzy1*(wabs*c2 + c3), MIN(c4, 0.5*MAX(wabs, c5))));
        if (wsize != 1.0) {
            double const wscale = 1.0/wsize;
            if (wsize > 1.0) *scale2 = (MAX(ascale, bsize)*wscale)*MIN(ascale, bsize);
            else *scale2 = (MIN(ascale, bsize)*wscale)*MAX(ascale, bsize);
            *wr2 *= wscale;
        } else *scale2 = ascale*bsize;
    }
}

/* The eigenvalues (real, imag, beta) of a generalized real Schur form read off its diagonal blocks, as
 * the reference's test hooks do (test/common/checks.c:59-100 eigenvalue_crawler): a 1 x 1 block gives
 * (S(i,i), 0, T(i,i)), a 2 x 2 block goes through DLAG2 (compute_complex_eigenvalue, common/math.c:148-176:
 * real = wr, imag = +-wi, beta = scale). */
void oracle_gep_extract_eigenvalues(int n, double const *S, int lds, double const *T, int ldt,
    double *ar, double *ai, double *be)
{
    int k = 0;
    while (k < n) {
        if (k+1 < n && EL(S,lds,k+1,k) != 0.0) {
            double s1, s2, w1, w2, wi;
            oracle_dlag2(&EL(S,lds,k,k), lds, &EL(T,ldt,k,k), ldt, DBL_MIN, &s1, &s2, &w1, &w2, &wi);
            ar[k] = w1; ai[k] = wi; be[k] = s1;
            ar[k+1] = w2; ai[k+1] = -wi; be[k+1] = s2;
            k += 2;
        } else {
            ar[k] = EL(S,lds,k,k); ai[k] = 0.0; be[k] = EL(T,ldt,k,k);
            k++;
        }
    }
}

/* Generalized real Schur form of the Hessenberg-triangular pencil (A, B); Q <- Q U1,
 * Z <- Z U2.  Returns 0, or the number of unconverged rows. */
int oracle_gep_schur(int n, double *A, int lda, double *B, int ldb, double *Q, int ldq,
    double *Z, int ldz, double *ar, double *ai, double *be)
{
    double const ulp = DBL_EPSILON, safmin = DBL_MIN;
    double bnorm = 0.0;
    for (int j = 0; j < n; j++) for (int i = 0; i <= j; i++) bnorm += EL(B,ldb,i,j)*EL(B,ldb,i,j);
    bnorm = sqrt(bnorm);
    double const btol = MAX(safmin, ulp*bnorm);
    int ihi = n-1, iter = 0, since = 0;
    int const maxit = 30*n;
    while (ihi >= 0) {
        if (iter > maxit) return ihi+1;
        /* numerically singular B: perturb (the same policy the device path documents) */
        if (fabs(EL(B,ldb,ihi,ihi)) < btol) EL(B,ldb,ihi,ihi) = (EL(B,ldb,ihi,ihi) < 0.0) ? -btol : btol;
        int ilo = ihi;
        while (ilo > 0) {
            double h = fabs(EL(A,lda,ilo,ilo-1));
            if (h <= MAX(safmin, ulp*(fabs(EL(A,lda,ilo-1,ilo-1)) + fabs(EL(A,lda,ilo,ilo))))) {
                EL(A,lda,ilo,ilo-1) = 0.0; break;
            }
            if (fabs(EL(B,ldb,ilo-1,ilo-1)) < btol)
                EL(B,ldb,ilo-1,ilo-1) = (EL(B,ldb,ilo-1,ilo-1) < 0.0) ? -btol : btol;
            ilo--;
        }
        if (ilo == ihi) {
            /* 1x1: beta >= 0 (dhgeqz flips the column) */
            if (EL(B,ldb,ihi,ihi) < 0.0) {
                for (int i = 0; i <= ihi; i++) { EL(A,lda,i,ihi) = -EL(A,lda,i,ihi); EL(B,ldb,i,ihi) = -EL(B,ldb,i,ihi); }
                for (int i = 0; i < n; i++) EL(Z,ldz,i,ihi) = -EL(Z,ldz,i,ihi);
            }
            ihi--; since = 0; continue;
        }
        if (ilo == ihi-1) {
            if (standardise(n, ilo, A, lda, B, ldb, Q, ldq, Z, ldz)) continue;   /* split: 1x1s next */
            ihi -= 2; since = 0; continue;
        }
        iter++; since++;
        /* ---- one implicit double-shift QZ step on [ilo, ihi] -------------------------------- */
        double sum, prod;
        {   /* shifts: eigenvalues of the trailing 2x2 of A B^-1 (needs the trailing 3x3 of B^-1) */
            int m = ihi;
            double b00 = EL(B,ldb,m-2,m-2), b01 = EL(B,ldb,m-2,m-1), b02 = EL(B,ldb,m-2,m);
            double b11 = EL(B,ldb,m-1,m-1), b12 = EL(B,ldb,m-1,m), b22 = EL(B,ldb,m,m);
            /* columns 1,2 of inv(B3) */
            double i11 = 1.0/b11, i01 = -b01*i11/b00;
            double i22 = 1.0/b22, i12 = -b12*i22/b11, i02 = -(b01*i12 + b02*i22)/b00;
            double a10 = EL(A,lda,m-1,m-2), a11 = EL(A,lda,m-1,m-1), a12 = EL(A,lda,m-1,m);
            double a21 = EL(A,lda,m,m-1), a22 = EL(A,lda,m,m);
            double m11 = a10*i01 + a11*i11, m12 = a10*i02 + a11*i12 + a12*i22;
            double m21 = a21*i11, m22 = a21*i12 + a22*i22;
            sum = m11 + m22; prod = m11*m22 - m12*m21;
            if (since % 10 == 0) {   /* exceptional shift */
                double e = fabs(EL(A,lda,m,m-1)/b11) + fabs(EL(A,lda,m-1,m-2)/b00);
                sum = 1.5*e; prod = e*e;
            }
        }
        double x[3];
        {
            int k = ilo;
            double b00 = EL(B,ldb,k,k), b01 = EL(B,ldb,k,k+1), b11 = EL(B,ldb,k+1,k+1);
            double a00 = EL(A,lda,k,k), a10 = EL(A,lda,k+1,k), a01 = EL(A,lda,k,k+1), a11 = EL(A,lda,k+1,k+1);
            double a21 = EL(A,lda,k+2,k+1);
            double z0 = a00/b00, z1 = a10/b00;
            double t1 = z1/b11, t0 = (z0 - b01*t1)/b00;
            x[0] = a00*t0 + a01*t1 - sum*z0 + prod;
            x[1] = a10*t0 + a11*t1 - sum*z1;
            x[2] = a21*t1;
        }
        for (int k = ilo; k <= ihi-2; k++) {
            double v[3], tau;
            house(3, x, v, &tau);
            int const c0 = (k > ilo) ? k-1 : k;
            apply_left(3, v, tau, A, lda, k, c0, n);
            apply_left(3, v, tau, B, ldb, k, k, n);
            apply_right(3, v, tau, Q, ldq, k, 0, n);
            if (k > ilo) { EL(A,lda,k+1,k-1) = 0.0; EL(A,lda,k+2,k-1) = 0.0; }
            /* Z_k1: zero B(k+2,k), B(k+2,k+1) */
            double y[3] = {EL(B,ldb,k+2,k), EL(B,ldb,k+2,k+1), EL(B,ldb,k+2,k+2)};
            house_last(3, y, v, &tau);
            /* v is normalised on its LAST entry by house_last (v[2] = 1) */
            int const rA = MIN(k+4, ihi+1);
            apply_right(3, v, tau, A, lda, k, 0, rA);
            apply_right(3, v, tau, B, ldb, k, 0, k+3);
            apply_right(3, v, tau, Z, ldz, k, 0, n);
            EL(B,ldb,k+2,k) = 0.0; EL(B,ldb,k+2,k+1) = 0.0;
            /* Z_k2: zero B(k+1,k) */
            double y2[2] = {EL(B,ldb,k+1,k), EL(B,ldb,k+1,k+1)};
            house_last(2, y2, v, &tau);
            apply_right(2, v, tau, A, lda, k, 0, rA);
            apply_right(2, v, tau, B, ldb, k, 0, k+2);
            apply_right(2, v, tau, Z, ldz, k, 0, n);
            EL(B,ldb,k+1,k) = 0.0;
            x[0] = EL(A,lda,k+1,k); x[1] = EL(A,lda,k+2,k);
            if (k < ihi-2) x[2] = EL(A,lda,k+3,k);
        }
        {
            int k = ihi-1;
            double v[3], tau;
            house(2, x, v, &tau);
            apply_left(2, v, tau, A, lda, k, k-1, n);
            apply_left(2, v, tau, B, ldb, k, k, n);
            apply_right(2, v, tau, Q, ldq, k, 0, n);
            EL(A,lda,k+1,k-1) = 0.0;
            double y2[2] = {EL(B,ldb,k+1,k), EL(B,ldb,k+1,k+1)};
            house_last(2, y2, v, &tau);
            apply_right(2, v, tau, A, lda, k, 0, ihi+1);
            apply_right(2, v, tau, B, ldb, k, 0, k+2);
            apply_right(2, v, tau, Z, ldz, k, 0, n);
            EL(B,ldb,k+1,k) = 0.0;
        }
    }
    oracle_gep_extract_eigenvalues(n, A, lda, B, ldb, ar, ai, be);
    return 0;
}

/* Structure of a generalized real Schur form: S quasi upper triangular without adjacent
 * 2x2 blocks sharing a row, T upper triangular; every 2x2 block has complex eigenvalues
 * (or a numerically double real one), a diagonal T block with positive entries.
 * Returns the number of violations. */
int oracle_check_gep_schur_form(int n, double const *S, int lds, double const *T, int ldt)
{
    int bad = 0;
    for (int j = 0; j < n; j++) {
        for (int i = j+2; i < n; i++) if (EL(S,lds,i,j) != 0.0) bad++;
        for (int i = j+1; i < n; i++) if (EL(T,ldt,i,j) != 0.0) bad++;
    }
    for (int k = 0; k+1 < n; k++) {
        if (EL(S,lds,k+1,k) == 0.0) continue;
        if (k+2 < n && EL(S,lds,k+2,k+1) != 0.0) bad++;
        double a11 = EL(S,lds,k,k), a12 = EL(S,lds,k,k+1), a21 = EL(S,lds,k+1,k), a22 = EL(S,lds,k+1,k+1);
        double b11 = EL(T,ldt,k,k), b12 = EL(T,ldt,k,k+1), b22 = EL(T,ldt,k+1,k+1);
        double p = b11*b22, q = a11*b22 + a22*b11 - a21*b12, r = a11*a22 - a12*a21;
        /* complex pair; a numerically double eigenvalue (relative gap below 3e-5) is on the
         * knife edge between the two classifications and accepted either way */
        if (q*q - 4.0*p*r > 1e-9*(q*q + fabs(4.0*p*r))) bad++;
        if (b12 != 0.0) bad++;
        if (!(b11 > 0.0) || !(b22 > 0.0)) bad++;
    }
    return bad;
}
