/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU oracle for the Schur leg.  The
 * reference reduces small segments with LAPACK dhseqr("S","V")
 * (src/schur/cpu_utils.c:2248-2309, call at :2292) and extracts eigenvalues
 * from the 1x1 / 2x2 diagonal blocks with LAPACK dlanv2
 * (src/common/math.c:148-195, src/schur/cpu_utils.c:3493-3520).  LAPACK is a
 * third-party dependency that is neither vendored nor version-pinned by the
 * reference, so this file restates the PUBLISHED LAPACK
 * algorithms those calls resolve to:
 *   - dlanv2  : standardisation of a real 2x2 block,
 *   - dlahqr  : the implicit double-shift QR iteration (what dhseqr runs for
 *               small matrices and the comparator `--solver lapack` of the
 *               reference test driver, test/schur/solvers.c:135-169),
 * written from the algorithm descriptions (Golub & Van Loan Alg. 7.5.1/7.5.2,
 * LAPACK Working Note 147 deflation criterion).  Any multi-shift/AED variant
 * must produce the same eigenvalues (as a multiset) and a valid real Schur
 * form; Schur forms themselves are only comparable through invariants.
 *
 * Pin: eigenvalues against tests/golden (numpy/LAPACK eigvals on the reference
 * test driver's LCG matrices) and the reference's own Schur checks
 * (test/common/hooks.c:535-714: quasi-triangular, 2x2 blocks standardised;
 * :891-991: returned eigenvalues vs eigenvalues of the diagonal blocks).
 * Bitwise parity is unpinned.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define MIN(a,b) ((a) < (b) ? (a) : (b))
#define MAX(a,b) ((a) > (b) ? (a) : (b))
#define SIGN(a,b) ((b) >= 0.0 ? fabs(a) : -fabs(a))

static double lapy2(double x, double y)
{
    double xa = fabs(x), ya = fabs(y);
    double w = MAX(xa, ya), z = MIN(xa, ya);
    if (z == 0.0) return w;
    return w*sqrt(1.0+(z/w)*(z/w));
}

/* LAPACK dlanv2: Schur factorisation of a real 2x2 block in standard form:
 * [a b; c d] = [cs -sn; sn cs] [aa bb; cc dd] [cs sn; -sn cs], with either
 * cc = 0 (two real eigenvalues) or aa = dd and bb*cc < 0 (complex pair). */
void oracle_dlanv2(double *a, double *b, double *c, double *d,
    double *rt1r, double *rt1i, double *rt2r, double *rt2i,
    double *cs, double *sn)
{
    double const eps = DBL_EPSILON*0.5;   /* dlamch('P')/base handling: 'P' = eps*base */
    double const multpl = 4.0;
    if (*c == 0.0) {
        *cs = 1.0; *sn = 0.0;
    }
    else if (*b == 0.0) {
        *cs = 0.0; *sn = 1.0;
        double temp = *d; *d = *a; *a = temp;
        *b = -*c; *c = 0.0;
    }
    else if ((*a-*d) == 0.0 && SIGN(1.0,*b) != SIGN(1.0,*c)) {
        *cs = 1.0; *sn = 0.0;
    }
    else {
        double temp = *a-*d;
        double p = 0.5*temp;
        double bcmax = MAX(fabs(*b), fabs(*c));
        double bcmis = MIN(fabs(*b), fabs(*c))*SIGN(1.0,*b)*SIGN(1.0,*c);
        double scale = MAX(fabs(p), bcmax);
        double z = (p/scale)*p + (bcmax/scale)*bcmis;
        if (z >= multpl*2.0*eps) {
            /* real eigenvalues */
            z = p + SIGN(sqrt(scale)*sqrt(z), p);
            *a = *d + z;
            *d = *d - (bcmax/z)*bcmis;
            double tau = lapy2(*c, z);
            *cs = z/tau; *sn = *c/tau;
            *b = *b-*c; *c = 0.0;
        }
        else {
            /* complex, or real almost equal, eigenvalues: make diagonal equal */
            double sigma = *b+*c;
            double tau = lapy2(sigma, temp);
            *cs = sqrt(0.5*(1.0+fabs(sigma)/tau));
            *sn = -(p/(tau*(*cs)))*SIGN(1.0, sigma);
            double aa = *a*(*cs) + *b*(*sn), bb = -*a*(*sn) + *b*(*cs);
            double cc = *c*(*cs) + *d*(*sn), dd = -*c*(*sn) + *d*(*cs);
            *a = aa*(*cs) + cc*(*sn); *b = bb*(*cs) + dd*(*sn);
            *c = -aa*(*sn) + cc*(*cs); *d = -bb*(*sn) + dd*(*cs);
            temp = 0.5*(*a+*d);
            *a = temp; *d = temp;
            if (*c != 0.0) {
                if (*b != 0.0) {
                    if (SIGN(1.0,*b) == SIGN(1.0,*c)) {
                        /* real eigenvalues: reduce to upper triangular form */
                        double sab = sqrt(fabs(*b)), sac = sqrt(fabs(*c));
                        p = SIGN(sab*sac, *c);
                        tau = 1.0/sqrt(fabs(*b+*c));
                        *a = temp+p; *d = temp-p;
                        *b = *b-*c; *c = 0.0;
                        double cs1 = sab*tau, sn1 = sac*tau;
                        temp = *cs*cs1 - *sn*sn1;
                        *sn = *cs*sn1 + *sn*cs1;
                        *cs = temp;
                    }
                }
                else {
                    *b = -*c; *c = 0.0;
                    temp = *cs; *cs = -*sn; *sn = temp;
                }
            }
        }
    }
    *rt1r = *a; *rt2r = *d;
    if (*c == 0.0) { *rt1i = 0.0; *rt2i = 0.0; }
    else { *rt1i = sqrt(fabs(*b))*sqrt(fabs(*c)); *rt2i = -*rt1i; }
}

/* 3-element (or shorter) Householder: LAPACK dlarfg without the safmin loop */
static double larfg_small(int n, double *alpha, double *x)
{
    double xnorm = 0.0;
    for (int i = 0; i < n-1; i++) xnorm = lapy2(xnorm, x[i]);
    if (xnorm == 0.0) return 0.0;
    double beta = -SIGN(lapy2(*alpha, xnorm), *alpha);
    double tau = (beta-*alpha)/beta;
    double s = 1.0/(*alpha-beta);
    for (int i = 0; i < n-1; i++) x[i] *= s;
    *alpha = beta;
    return tau;
}

#define H_(i,j) H[(size_t)(j)*ldH+(i)]
#define Z_(i,j) Z[(size_t)(j)*ldZ+(i)]

/*
 * Implicit double-shift QR (LAPACK dlahqr, wantt = wantz = true) on the whole
 * n x n upper Hessenberg matrix H; Z <- Z*U.  real/imag receive the
 * eigenvalues in diagonal order.  Returns 0 on success, i+1 if the iteration
 * limit was hit while working on row i.
 */
int oracle_schur(int n, double *H, int ldH, double *Z, int ldZ,
    double *wr, double *wi)
{
    double const ulp = DBL_EPSILON;           /* dlamch('P') */
    double const safmin = DBL_MIN;
    double const smlnum = safmin*((double)n/ulp);
    int const itmax = 30*MAX(10, n);
    int const kexsh = 10;

    if (n == 0) return 0;
    /* clear out the trash below the sub-diagonal */
    for (int j = 0; j < n-2; j++) {
        H_(j+2,j) = 0.0;
        if (j+3 < n) H_(j+3,j) = 0.0;
    }

    int i = n-1;
    while (i >= 0) {
        int l = 0, its;
        int converged = 0;
        for (its = 0; its <= itmax; its++) {
            /* look for a single small sub-diagonal element */
            int k;
            for (k = i; k > l; k--) {
                if (fabs(H_(k,k-1)) <= smlnum) break;
                double tst = fabs(H_(k-1,k-1)) + fabs(H_(k,k));
                if (tst == 0.0) {
                    if (k-2 >= 0) tst += fabs(H_(k-1,k-2));
                    if (k+1 <= n-1) tst += fabs(H_(k+1,k));
                }
                if (fabs(H_(k,k-1)) <= ulp*tst) {
                    double ab = MAX(fabs(H_(k,k-1)), fabs(H_(k-1,k)));
                    double ba = MIN(fabs(H_(k,k-1)), fabs(H_(k-1,k)));
                    double aa = MAX(fabs(H_(k,k)), fabs(H_(k-1,k-1)-H_(k,k)));
                    double bb = MIN(fabs(H_(k,k)), fabs(H_(k-1,k-1)-H_(k,k)));
                    double s = aa+ab;
                    if (ba*(ab/s) <= MAX(smlnum, ulp*(bb*(aa/s)))) break;
                }
            }
            l = k;
            if (l > 0) H_(l,l-1) = 0.0;
            if (l >= i-1) { converged = 1; break; }

            /* shifts */
            double h11, h21, h12, h22;
            if (its % (2*kexsh) == 0 && its > 0) {
                double s = fabs(H_(l+1,l)) + fabs(H_(l+2,l+1));
                h11 = 0.75*s + H_(l,l); h12 = -0.4375*s; h21 = s; h22 = h11;
            }
            else if (its % kexsh == 0 && its > 0) {
                double s = fabs(H_(i,i-1)) + fabs(H_(i-1,i-2));
                h11 = 0.75*s + H_(i,i); h12 = -0.4375*s; h21 = s; h22 = h11;
            }
            else {
                h11 = H_(i-1,i-1); h21 = H_(i,i-1); h12 = H_(i-1,i); h22 = H_(i,i);
            }
            double rt1r, rt1i, rt2r, rt2i;
            double s = fabs(h11)+fabs(h12)+fabs(h21)+fabs(h22);
            if (s == 0.0) { rt1r = rt1i = rt2r = rt2i = 0.0; }
            else {
                h11 /= s; h21 /= s; h12 /= s; h22 /= s;
                double tr = (h11+h22)/2.0;
                double det = (h11-tr)*(h22-tr) - h12*h21;
                double rtdisc = sqrt(fabs(det));
                if (det >= 0.0) {
                    rt1r = tr*s; rt2r = rt1r; rt1i = rtdisc*s; rt2i = -rt1i;
                }
                else {
                    rt1r = tr+rtdisc; rt2r = tr-rtdisc;
                    if (fabs(rt1r-h22) <= fabs(rt2r-h22)) { rt1r *= s; rt2r = rt1r; }
                    else { rt2r *= s; rt1r = rt2r; }
                    rt1i = rt2i = 0.0;
                }
            }

            /* look for two consecutive small sub-diagonal elements */
            double v[3];
            int m;
            for (m = i-2; m >= l; m--) {
                double h21s = fabs(H_(m+1,m));
                double ss = fabs(H_(m,m)-rt2r) + fabs(rt2i) + h21s;
                h21s = H_(m+1,m)/ss;
                v[0] = h21s*H_(m,m+1) + (H_(m,m)-rt1r)*((H_(m,m)-rt2r)/ss) - rt1i*(rt2i/ss);
                v[1] = h21s*(H_(m,m)+H_(m+1,m+1)-rt1r-rt2r);
                v[2] = h21s*H_(m+2,m+1);
                ss = fabs(v[0])+fabs(v[1])+fabs(v[2]);
                v[0] /= ss; v[1] /= ss; v[2] /= ss;
                if (m == l) break;
                double h00 = fabs(H_(m-1,m-1)), hmm = fabs(H_(m,m)), h11a = fabs(H_(m+1,m+1));
                if (fabs(H_(m,m-1))*(fabs(v[1])+fabs(v[2])) <= ulp*fabs(v[0])*(h00+hmm+h11a))
                    break;
            }

            /* double-shift QR step */
            for (int kk = m; kk <= i-1; kk++) {
                int nr = MIN(3, i-kk+1);
                if (kk > m) {
                    v[0] = H_(kk,kk-1); v[1] = H_(kk+1,kk-1);
                    if (nr == 3) v[2] = H_(kk+2,kk-1);
                }
                double t1 = larfg_small(nr, &v[0], &v[1]);
                if (kk > m) {
                    H_(kk,kk-1) = v[0]; H_(kk+1,kk-1) = 0.0;
                    if (kk < i-1) H_(kk+2,kk-1) = 0.0;
                }
                else if (m > l) {
                    H_(kk,kk-1) *= (1.0-t1);
                }
                double v2 = v[1], t2 = t1*v2;
                if (nr == 3) {
                    double v3 = v[2], t3 = t1*v3;
                    for (int j = kk; j < n; j++) {
                        double sum = H_(kk,j) + v2*H_(kk+1,j) + v3*H_(kk+2,j);
                        H_(kk,j) -= sum*t1; H_(kk+1,j) -= sum*t2; H_(kk+2,j) -= sum*t3;
                    }
                    int jend = MIN(kk+3, i);
                    for (int j = 0; j <= jend; j++) {
                        double sum = H_(j,kk) + v2*H_(j,kk+1) + v3*H_(j,kk+2);
                        H_(j,kk) -= sum*t1; H_(j,kk+1) -= sum*t2; H_(j,kk+2) -= sum*t3;
                    }
                    for (int j = 0; j < n; j++) {
                        double sum = Z_(j,kk) + v2*Z_(j,kk+1) + v3*Z_(j,kk+2);
                        Z_(j,kk) -= sum*t1; Z_(j,kk+1) -= sum*t2; Z_(j,kk+2) -= sum*t3;
                    }
                }
                else if (nr == 2) {
                    for (int j = kk; j < n; j++) {
                        double sum = H_(kk,j) + v2*H_(kk+1,j);
                        H_(kk,j) -= sum*t1; H_(kk+1,j) -= sum*t2;
                    }
                    for (int j = 0; j <= i; j++) {
                        double sum = H_(j,kk) + v2*H_(j,kk+1);
                        H_(j,kk) -= sum*t1; H_(j,kk+1) -= sum*t2;
                    }
                    for (int j = 0; j < n; j++) {
                        double sum = Z_(j,kk) + v2*Z_(j,kk+1);
                        Z_(j,kk) -= sum*t1; Z_(j,kk+1) -= sum*t2;
                    }
                }
            }
        }
        if (!converged) return i+1;

        if (l == i) {
            wr[i] = H_(i,i); wi[i] = 0.0;
        }
        else {   /* l == i-1: a 2x2 block */
            double cs, sn;
            oracle_dlanv2(&H_(i-1,i-1), &H_(i-1,i), &H_(i,i-1), &H_(i,i),
                &wr[i-1], &wi[i-1], &wr[i], &wi[i], &cs, &sn);
            for (int j = i+1; j < n; j++) {
                double x = H_(i-1,j), y = H_(i,j);
                H_(i-1,j) = cs*x + sn*y; H_(i,j) = cs*y - sn*x;
            }
            for (int j = 0; j < i-1; j++) {
                double x = H_(j,i-1), y = H_(j,i);
                H_(j,i-1) = cs*x + sn*y; H_(j,i) = cs*y - sn*x;
            }
            for (int j = 0; j < n; j++) {
                double x = Z_(j,i-1), y = Z_(j,i);
                Z_(j,i-1) = cs*x + sn*y; Z_(j,i) = cs*y - sn*x;
            }
        }
        i = l-1;
    }
    return 0;
}

/* Eigenvalues from the diagonal blocks of a quasi-triangular matrix
 * (src/schur/cpu_utils.c:3493-3520 + src/common/math.c:178-186). */
void oracle_extract_eigenvalues(int n, double const *S, int ldS,
    double *wr, double *wi)
{
    for (int i = 0; i < n; i++) {
        if (i+1 < n && S[(size_t)i*ldS+i+1] != 0.0) {
            double a = S[(size_t)i*ldS+i], b = S[(size_t)(i+1)*ldS+i];
            double c = S[(size_t)i*ldS+i+1], d = S[(size_t)(i+1)*ldS+i+1];
            double cs, sn;
            oracle_dlanv2(&a, &b, &c, &d, &wr[i], &wi[i], &wr[i+1], &wi[i+1], &cs, &sn);
            i++;
        }
        else {
            wr[i] = S[(size_t)i*ldS+i]; wi[i] = 0.0;
        }
    }
}

/*
 * Schur-form check of the reference test driver (test/common/hooks.c:535-714):
 * returns 0 if S is quasi-upper-triangular (exact zeros below the
 * sub-diagonal, no two consecutive non-zero sub-diagonal entries) and every
 * 2x2 block is in standard form (equal diagonal, off-diagonal of opposite
 * sign); otherwise a positive code: 1 = entry below sub-diagonal,
 * 2 = consecutive sub-diagonals, 3 = non-standard 2x2 block.
 */
int oracle_check_schur_form(int n, double const *S, int ldS)
{
    for (int j = 0; j < n; j++)
        for (int i = j+2; i < n; i++)
            if (S[(size_t)j*ldS+i] != 0.0) return 1;
    for (int i = 0; i+1 < n; i++) {
        double sub = S[(size_t)i*ldS+i+1];
        if (sub == 0.0) continue;
        if (i+2 < n && S[(size_t)(i+1)*ldS+i+2] != 0.0) return 2;
        double a = S[(size_t)i*ldS+i], d = S[(size_t)(i+1)*ldS+i+1];
        double b = S[(size_t)(i+1)*ldS+i];
        if (a != d) return 3;
        if (!(b*sub < 0.0)) return 3;
    }
    return 0;
}
