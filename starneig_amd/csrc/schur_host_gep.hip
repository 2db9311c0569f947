// Host-side sequential kernels of the GENERALIZED Schur path (QZ, row S9 of SURVEY.md 8a):
// small Hessenberg-triangular pencils that sit on the critical path of the multi-shift QZ
// iteration -- the AED window and the final small blocks.
//
// The reference runs these on a CPU worker through LAPACK (dhgeqz for pencils <= 64 rows,
// its own sequential QZ loop above that, dgghrd, dtgexc, dlagv2, dlag2:
// schur/cpu_utils.c:2248-2309, :2651-2716, :3185-3371, common/math.c:148-176).  No LAPACK
// here: the published algorithms (Moler & Stewart's QZ step as in Golub & Van Loan
// Alg. 7.7.2, the dgghrd Givens scheme) are written out for windows of a few hundred rows.
// Differences to LAPACK's dhgeqz: always the implicit double-shift step (real shift pairs
// included).  Infinite eigenvalues: a diagonal entry of B below the infinity threshold inside an
// active block is chased to the top of the block and deflated there (gep_push_inf_window: the
// rotation scheme of the reference's push_inf_up, schur/cpu_utils.c:360-425, driven window by
// window from schur_gep.hip); the small-pencil kernels below only guard against entries that
// turn up mid-iteration (perturbation to u*||B||_F, a backward error of the size of rounding).
// Reordering inside the AED window swaps adjacent blocks through the generalized
// Sylvester equation like LAPACK dtgex2.
#include "schur_host.h"
#include "tuning.h"
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cfloat>
#include <cstring>
#include <vector>
#include <algorithm>

namespace sn { namespace host {

namespace {

inline double sgn(double a, double b) { return b >= 0.0 ? std::fabs(a) : -std::fabs(a); }

// sqrt(a^2 + b^2): plain when both are far from the ends of the exponent range (always, in practice;
// hypot costs several times the rest of a rotation)
inline double norm2(double a, double b)
{
    double const x = std::fabs(a), y = std::fabs(b), hi = std::max(x, y), lo = std::min(x, y);
    if (hi < 1e150 && hi > 1e-150 && (lo > 1e-150 || lo == 0.0)) return std::sqrt(x * x + y * y);
    return std::hypot(a, b);
}

// Givens: [c s; -s c] [f; g] = [r; 0]
inline void givens(double f, double g, double &c, double &s, double &r)
{
    if (g == 0.0) { c = 1.0; s = 0.0; r = f; return; }
    if (f == 0.0) { c = 0.0; s = 1.0; r = g; return; }
    r = norm2(f, g); c = f / r; s = g / r;
}

struct Mat { double *p; int ld; inline double &operator()(int i, int j) const { return p[(size_t)j * ld + i]; } };

// rows r1,r2 over columns [c0,c1): x' = c x + s y, y' = c y - s x.  (Row walks on a column-major matrix
// do not vectorise; four columns are loaded before any is stored so that the compiler need not
// serialise them.)
inline void rot_rows(Mat M, int r1, int r2, int c0, int c1, double c, double s)
{
    int j = c0;
    size_t const ld = M.ld;
    for (; j + 4 <= c1; j += 4) {
        double *p0 = M.p + (size_t)j * ld, *p1 = p0 + ld, *p2 = p1 + ld, *p3 = p2 + ld;
        double x0 = p0[r1], y0 = p0[r2], x1 = p1[r1], y1 = p1[r2], x2 = p2[r1], y2 = p2[r2], x3 = p3[r1], y3 = p3[r2];
        p0[r1] = c * x0 + s * y0; p0[r2] = c * y0 - s * x0;
        p1[r1] = c * x1 + s * y1; p1[r2] = c * y1 - s * x1;
        p2[r1] = c * x2 + s * y2; p2[r2] = c * y2 - s * x2;
        p3[r1] = c * x3 + s * y3; p3[r2] = c * y3 - s * x3;
    }
    for (; j < c1; j++) { double x = M(r1, j), y = M(r2, j); M(r1, j) = c * x + s * y; M(r2, j) = c * y - s * x; }
}
inline void rot_cols(Mat M, int c1_, int c2_, int r0, int r1, double c, double s)
{
    double *__restrict__ a = &M(0, c1_), *__restrict__ b = &M(0, c2_);      // distinct columns: vectorises
    for (int i = r0; i < r1; i++) { double x = a[i], y = b[i]; a[i] = c * x + s * y; b[i] = c * y - s * x; }
}

// sqrt of the sum of squares of x[i0:i1): plain unless an entry is near the ends of the exponent range
inline double norm_range(const double *x, int i0, int i1)
{
    double hi = 0.0, lo = DBL_MAX, ssq = 0.0;
    for (int i = i0; i < i1; i++) { double a = std::fabs(x[i]); hi = std::max(hi, a); if (a != 0.0) lo = std::min(lo, a); ssq += a * a; }
    if (hi == 0.0) return 0.0;
    if (hi < 1e150 && lo > 1e-150) return std::sqrt(ssq);
    double xn = 0.0;
    for (int i = i0; i < i1; i++) xn = std::hypot(xn, x[i]);
    return xn;
}
// Householder with the pivot FIRST: (I - tau v v^T) x = beta e_1, v[0] = 1
inline double house_first(int n, const double *x, double *v, double &beta)
{
    double xn = norm_range(x, 1, n);
    v[0] = 1.0;
    if (xn == 0.0) { for (int i = 1; i < n; i++) v[i] = 0.0; beta = x[0]; return 0.0; }
    beta = -sgn(norm2(x[0], xn), x[0]);
    double sc = 1.0 / (x[0] - beta);
    for (int i = 1; i < n; i++) v[i] = x[i] * sc;
    return (beta - x[0]) / beta;
}
// Householder with the pivot LAST: x^T (I - tau v v^T) = beta e_n^T, v[n-1] = 1
inline double house_last(int n, const double *x, double *v, double &beta)
{
    double xn = norm_range(x, 0, n - 1);
    v[n - 1] = 1.0;
    if (xn == 0.0) { for (int i = 0; i + 1 < n; i++) v[i] = 0.0; beta = x[n - 1]; return 0.0; }
    beta = -sgn(norm2(x[n - 1], xn), x[n - 1]);
    double sc = 1.0 / (x[n - 1] - beta);
    for (int i = 0; i + 1 < n; i++) v[i] = x[i] * sc;
    return (beta - x[n - 1]) / beta;
}
// M(r0:r0+len, c0:c1) <- (I - tau v v^T) M
inline void refl_left(Mat M, int r0, int len, int c0, int c1, const double *v, double tau)
{
    if (tau == 0.0) return;
    if (len == 3) {
        double const v0 = v[0], v1 = v[1], v2 = v[2], t0 = tau * v0, t1 = tau * v1, t2 = tau * v2;
        size_t const ld = M.ld;
        int j = c0;
        for (; j + 4 <= c1; j += 4) {
            double *p0 = M.p + (size_t)j * ld + r0, *p1 = p0 + ld, *p2 = p1 + ld, *p3 = p2 + ld;
            double a0 = p0[0], a1 = p0[1], a2 = p0[2], b0 = p1[0], b1 = p1[1], b2 = p1[2];
            double e0 = p2[0], e1 = p2[1], e2 = p2[2], d0 = p3[0], d1 = p3[1], d2 = p3[2];
            double sa = v0 * a0 + v1 * a1 + v2 * a2, sb = v0 * b0 + v1 * b1 + v2 * b2;
            double se = v0 * e0 + v1 * e1 + v2 * e2, sd = v0 * d0 + v1 * d1 + v2 * d2;
            p0[0] = a0 - sa * t0; p0[1] = a1 - sa * t1; p0[2] = a2 - sa * t2;
            p1[0] = b0 - sb * t0; p1[1] = b1 - sb * t1; p1[2] = b2 - sb * t2;
            p2[0] = e0 - se * t0; p2[1] = e1 - se * t1; p2[2] = e2 - se * t2;
            p3[0] = d0 - sd * t0; p3[1] = d1 - sd * t1; p3[2] = d2 - sd * t2;
        }
        for (; j < c1; j++) {
            double *p = M.p + (size_t)j * ld + r0;
            double a0 = p[0], a1 = p[1], a2 = p[2], sa = v0 * a0 + v1 * a1 + v2 * a2;
            p[0] = a0 - sa * t0; p[1] = a1 - sa * t1; p[2] = a2 - sa * t2;
        }
        return;
    }
    for (int j = c0; j < c1; j++) {
        double *__restrict__ x = &M(r0, j);
        double s = 0.0;
        for (int i = 0; i < len; i++) s += v[i] * x[i];
        s *= tau;
        for (int i = 0; i < len; i++) x[i] -= s * v[i];
    }
}
// M(r0:r1, c0:c0+len) <- M (I - tau v v^T): column sweeps over contiguous rows
inline void refl_right(Mat M, int c0, int len, int r0, int r1, const double *v, double tau)
{
    if (tau == 0.0 || r1 <= r0) return;
    if (len == 3) {
        double *__restrict__ x0 = &M(0, c0), *__restrict__ x1 = &M(0, c0 + 1), *__restrict__ x2 = &M(0, c0 + 2);
        double const v0 = v[0], v1 = v[1], v2 = v[2], t0 = tau * v0, t1 = tau * v1, t2 = tau * v2;
        for (int i = r0; i < r1; i++) {
            double s = x0[i] * v0 + x1[i] * v1 + x2[i] * v2;
            x0[i] -= s * t0; x1[i] -= s * t1; x2[i] -= s * t2;
        }
        return;
    }
    if (len == 2) {
        double *__restrict__ x0 = &M(0, c0), *__restrict__ x1 = &M(0, c0 + 1);
        double const v0 = v[0], v1 = v[1], t0 = tau * v0, t1 = tau * v1;
        for (int i = r0; i < r1; i++) {
            double s = x0[i] * v0 + x1[i] * v1;
            x0[i] -= s * t0; x1[i] -= s * t1;
        }
        return;
    }
    if (len == 4) {
        double *__restrict__ x0 = &M(0, c0), *__restrict__ x1 = &M(0, c0 + 1), *__restrict__ x2 = &M(0, c0 + 2), *__restrict__ x3 = &M(0, c0 + 3);
        double const v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3], t0 = tau * v0, t1 = tau * v1, t2 = tau * v2, t3 = tau * v3;
        for (int i = r0; i < r1; i++) {
            double s = x0[i] * v0 + x1[i] * v1 + x2[i] * v2 + x3[i] * v3;
            x0[i] -= s * t0; x1[i] -= s * t1; x2[i] -= s * t2; x3[i] -= s * t3;
        }
        return;
    }
    thread_local std::vector<double> w;
    if ((int)w.size() < r1) w.resize(r1);
    double *__restrict__ ww = w.data();
    for (int i = r0; i < r1; i++) ww[i] = 0.0;
    for (int j = 0; j < len; j++) {
        double const *__restrict__ x = &M(0, c0 + j); double const vj = v[j];
        for (int i = r0; i < r1; i++) ww[i] += x[i] * vj;
    }
    for (int j = 0; j < len; j++) {
        double *__restrict__ x = &M(0, c0 + j); double const tv = tau * v[j];
        for (int i = r0; i < r1; i++) x[i] -= ww[i] * tv;
    }
}

// Sum and product of the eigenvalues of the 2x2 pencil (a, b) with b upper triangular:
// b11 b22 l^2 - (a11 b22 + a22 b11 - a21 b12) l + (a11 a22 - a12 a21) = 0
inline void pencil2_sum_prod(double a11, double a12, double a21, double a22,
    double b11, double b12, double b22, double &sum, double &prod)
{
    double den = b11 * b22;
    sum = (a11 * b22 + a22 * b11 - a21 * b12) / den;
    prod = (a11 * a22 - a12 * a21) / den;
}

// Eigenvalues of the 2x2 pencil (a, b), b upper triangular with b11, b22 != 0, computed as
// shift + (pp +- sqrt(pp^2 + qq)) with shift = the smaller of a11/b11, a22/b22 (the published
// algorithm of LAPACK dlag2, which the reference calls: common/math.c:148-176).  The quadratic
// formula on (sum, prod) loses the difference of two close eigenvalues to cancellation -- a real
// pair -4423.50 / -4424.03 came out as a "complex" pair with zero imaginary part, an error of 1e-4.
// Returns true for a real pair: lr[0] the eigenvalue of larger magnitude, lr[1] the smaller one;
// false for a complex pair lr[0] +- i lr[1] (lr[1] > 0).
inline bool pencil2_eigenvalues(double a11, double a12, double a21, double a22,
    double b11, double b12, double b22, double lr[2])
{
    double const binv11 = 1.0 / b11, binv22 = 1.0 / b22;
    double const s1 = a11 * binv11, s2 = a22 * binv22;
    double const ss = a21 * (binv11 * binv22);
    double as12, pp, shift;
    if (std::fabs(s1) <= std::fabs(s2)) {
        as12 = a12 - s1 * b12;
        double const as22 = a22 - s1 * b22, abi22 = as22 * binv22 - ss * b12;
        pp = 0.5 * abi22; shift = s1;
    } else {
        as12 = a12 - s2 * b12;
        double const as11 = a11 - s2 * b11, abi22 = -ss * b12;
        pp = 0.5 * (as11 * binv11 + abi22); shift = s2;
    }
    double const qq = ss * as12, discr = pp * pp + qq;
    if (discr >= 0.0) {
        double const r = std::sqrt(discr), sr = (pp >= 0.0) ? r : -r;
        double wbig = shift + (pp + sr), wsmall = shift + (pp - sr);
        if (std::fabs(wbig) < std::fabs(wsmall)) std::swap(wbig, wsmall);
        if (0.5 * std::fabs(wbig) > std::max(std::fabs(wsmall), DBL_MIN))
            wsmall = ((a11 * a22 - a12 * a21) * (binv11 * binv22)) / wbig;
        lr[0] = wbig; lr[1] = wsmall;
        return true;
    }
    lr[0] = shift + pp; lr[1] = std::sqrt(-discr);
    return false;
}

// (alpha_r + i alpha_i, beta) of the two eigenvalues of a 2 x 2 pencil as the reference RETURNS them:
// LAPACK dlag2's outputs real = wr, imag = +-wi, beta = scale (common/math.c:148-176; dlag2 restated from
// its published algorithm, scalings included).  The test hook evaluates the same routine on the same
// block (test/common/checks.c:82), and on a block whose discriminant nearly vanishes one rounding moves
// the pair by sqrt(u): no floating-point contraction here, so that this function, the test suite's own
// restatement and LAPACK itself give the same bits (tests/golden/dlag2_cases.npz).
void pencil2_dlag2(double const *A, int lda, double const *B, int ldb,
    double &scale1, double &scale2, double &wr1, double &wr2, double &wi)
{
#pragma clang fp contract(off)
    double const safmin = DBL_MIN, fuzzy1 = 1.0 + 1.0e-5;
    double const rtmin = std::sqrt(safmin), rtmax = 1.0 / rtmin, safmax = 1.0 / safmin;
    auto mx = [](double a, double b) { return a > b ? a : b; };
    auto mn = [](double a, double b) { return a < b ? a : b; };
    double const anorm = mx(mx(std::fabs(A[0]) + std::fabs(A[1]), std::fabs(A[lda]) + std::fabs(A[lda + 1])), safmin);
    double const ascale = 1.0 / anorm;
    double const a11 = ascale * A[0], a21 = ascale * A[1], a12 = ascale * A[lda], a22 = ascale * A[lda + 1];
    double b11 = B[0], b12 = B[ldb], b22 = B[ldb + 1];
    double const bmin = rtmin * mx(mx(std::fabs(b11), std::fabs(b12)), mx(std::fabs(b22), rtmin));
    if (std::fabs(b11) < bmin) b11 = std::copysign(bmin, b11);
    if (std::fabs(b22) < bmin) b22 = std::copysign(bmin, b22);
    double const bnorm = mx(mx(std::fabs(b11), std::fabs(b12) + std::fabs(b22)), safmin);
    double const bsize = mx(std::fabs(b11), std::fabs(b22));
    double const bscale = 1.0 / bsize;
    b11 *= bscale; b12 *= bscale; b22 *= bscale;
    double const binv11 = 1.0 / b11, binv22 = 1.0 / b22;
    double const s1 = a11 * binv11, s2 = a22 * binv22;
    double as12, abi22, pp, shift, ss;
    if (std::fabs(s1) <= std::fabs(s2)) {
        as12 = a12 - s1 * b12;
        double const as22 = a22 - s1 * b22;
        ss = a21 * (binv11 * binv22);
        abi22 = as22 * binv22 - ss * b12;
        pp = 0.5 * abi22;
        shift = s1;
    } else {
        as12 = a12 - s2 * b12;
        double const as11 = a11 - s2 * b11;
        ss = a21 * (binv11 * binv22);
        abi22 = -ss * b12;
        pp = 0.5 * (as11 * binv11 + abi22);
        shift = s2;
    }
    double const qq = ss * as12;
    double discr, r;
    if (std::fabs(pp * rtmin) >= 1.0) {
        double const t = rtmin * pp;
        discr = t * t + qq * safmin;
        r = std::sqrt(std::fabs(discr)) * rtmax;
    } else if (pp * pp + std::fabs(qq) <= safmin) {
        double const t = rtmax * pp;
        discr = t * t + qq * safmax;
        r = std::sqrt(std::fabs(discr)) * rtmin;
    } else {
        discr = pp * pp + qq;
        r = std::sqrt(std::fabs(discr));
    }
    if (discr >= 0.0 || r == 0.0) {
        double const sr = std::copysign(r, pp);
        double const sum = pp + sr, diff = pp - sr;
        double const wbig = shift + sum;
        double wsmall = shift + diff;
        if (0.5 * std::fabs(wbig) > mx(std::fabs(wsmall), safmin)) {
            double const wdet = (a11 * a22 - a12 * a21) * (binv11 * binv22);
            wsmall = wdet / wbig;
        }
        if (pp > abi22) { wr1 = mn(wbig, wsmall); wr2 = mx(wbig, wsmall); }
        else { wr1 = mx(wbig, wsmall); wr2 = mn(wbig, wsmall); }
        wi = 0.0;
    } else {
        wr1 = shift + pp; wr2 = wr1; wi = r;
    }
    double const c1 = bsize * (safmin * mx(1.0, ascale));
    double const c2 = safmin * mx(1.0, bnorm);
    double const c3 = bsize * safmin;
    double const c4 = (ascale <= 1.0 && bsize <= 1.0) ? mn(1.0, (ascale / safmin) * bsize) : 1.0;
    double const c5 = (ascale <= 1.0 || bsize <= 1.0) ? mn(1.0, ascale * bsize) : 1.0;
    double wabs = std::fabs(wr1) + std::fabs(wi);
    double wsize = mx(mx(safmin, c1), mx(fuzzy1 * (wabs * c2 + c3), mn(c4, 0.5 * mx(wabs, c5))));
    if (wsize != 1.0) {
        double const wscale = 1.0 / wsize;
        if (wsize > 1.0) scale1 = (mx(ascale, bsize) * wscale) * mn(ascale, bsize);
        else scale1 = (mn(ascale, bsize) * wscale) * mx(ascale, bsize);
        wr1 *= wscale;
        if (wi != 0.0) { wi *= wscale; wr2 = wr1; scale2 = scale1; }
    } else {
        scale1 = ascale * bsize;
        scale2 = scale1;
    }
    if (wi == 0.0) {
        wabs = std::fabs(wr2);
        wsize = mx(mx(safmin, c1), mx(fuzzy1 * (wabs * c2 + c3), mn(c4, 0.5 * mx(wabs, c5))));
        if (wsize != 1.0) {
            double const wscale = 1.0 / wsize;
            if (wsize > 1.0) scale2 = (mx(ascale, bsize) * wscale) * mn(ascale, bsize);
            else scale2 = (mn(ascale, bsize) * wscale) * mx(ascale, bsize);
            wr2 *= wscale;
        } else scale2 = ascale * bsize;
    }
}

} // namespace

// Standardises the 2x2 diagonal block at p of the pencil (A,B) (B upper triangular):
// real eigenvalues -> both A and B become upper triangular (returns 1);
// complex pair -> B block diagonal with positive entries, A block full (returns 2).
// (LAPACK dlagv2; reference process_2x2_block, cpu_utils.c:801-850.)
static int gep_standardise_2x2(int n, Mat A, Mat B, Mat Q, Mat Z, int nq, int p)
{
    // Real pair: one pass rotates an (approximate) eigenvector of the block to e1.  When the
    // two eigenvalues (almost) coincide the eigenvalue from the quadratic formula carries an
    // error of sqrt(eps) and so does the entry that should vanish; every pass is an orthogonal
    // transformation of the whole pencil, so passes are simply repeated (each one is an
    // exact-shift QZ step on the block; convergence is quadratic for distinct eigenvalues and
    // linear with ratio ~1/2 for a double one, hence the generous limit) until A(p+1,p) meets
    // the deflation criterion -- only then is it set to zero.
    double prev = HUGE_VAL;
    for (int pass = 0; pass < 60; pass++) {
        double a11 = A(p, p), a12 = A(p, p + 1), a21 = A(p + 1, p), a22 = A(p + 1, p + 1);
        double b11 = B(p, p), b12 = B(p, p + 1), b22 = B(p + 1, p + 1);
        double const tol = DBL_EPSILON * (std::fabs(a11) + std::fabs(a22));
        if (std::fabs(a21) <= tol) { A(p + 1, p) = 0.0; return 1; }
        if (std::fabs(a21) >= 0.25 * prev) {
            // stagnation at the rounding floor of the block (a few ulp of its largest entry)
            if (std::fabs(a21) <= 1000.0 * DBL_EPSILON * (std::fabs(a11) + std::fabs(a12) + std::fabs(a22))) {
                A(p + 1, p) = 0.0; return 1;
            }
            if (pass > 8) break;
        }
        prev = std::fabs(a21);
        double lr[2];
        if (!pencil2_eigenvalues(a11, a12, a21, a22, b11, b12, b22, lr)) break;
        // real pair: deflate the eigenvalue of SMALLER magnitude.  With a nearly singular B block
        // the other one is ~1/u and A - lam B would be dominated by lam B, losing the null vector
        double const lam = lr[1];
        // right null vector x of M = A22 - lam B22
        double m00 = a11 - lam * b11, m01 = a12 - lam * b12, m10 = a21, m11 = a22 - lam * b22;
        double x0, x1;
        if (std::hypot(m00, m01) >= std::hypot(m10, m11)) { x0 = m01; x1 = -m00; }
        else { x0 = m11; x1 = -m10; }
        double c, s, r;
        givens(x0, x1, c, s, r);                  // [c s; -s c][x0; x1] = [r; 0]
        // V = [c -s; s c] has first column x/|x|: apply M <- M V  (cols p, p+1)
        rot_cols(A, p, p + 1, 0, std::min(n, p + 2), c, s);
        rot_cols(B, p, p + 1, 0, p + 2, c, s);
        rot_cols(Z, p, p + 1, 0, nq, c, s);
        // rotate rows to annihilate B(p+1,p) (A(p+1,p) follows: A V e1 = lam B V e1)
        double c2, s2, r2;
        givens(B(p, p), B(p + 1, p), c2, s2, r2);
        rot_rows(A, p, p + 1, p, n, c2, s2);
        rot_rows(B, p, p + 1, p, n, c2, s2);
        rot_cols(Q, p, p + 1, 0, nq, c2, s2);
        B(p + 1, p) = 0.0;
    }
    double a11 = A(p, p), a12 = A(p, p + 1), a21 = A(p + 1, p), a22 = A(p + 1, p + 1);
    double b11 = B(p, p), b12 = B(p, p + 1), b22 = B(p + 1, p + 1);
    (void)a11; (void)a12; (void)a21; (void)a22;
    // complex pair: B22 = U S V^T  ->  U^T B22 V diagonal (2x2 SVD of a triangular matrix)
    double th = 0.5 * std::atan2(2.0 * b11 * b12, b11 * b11 - b12 * b12 - b22 * b22);
    double cv = std::cos(th), sv = std::sin(th);          // V = [cv -sv; sv cv]
    rot_cols(A, p, p + 1, 0, std::min(n, p + 2), cv, sv);
    rot_cols(B, p, p + 1, 0, p + 2, cv, sv);
    rot_cols(Z, p, p + 1, 0, nq, cv, sv);
    double c2, s2, r2;
    givens(B(p, p), B(p + 1, p), c2, s2, r2);             // columns of B22 V are orthogonal
    rot_rows(A, p, p + 1, p, n, c2, s2);
    rot_rows(B, p, p + 1, p, n, c2, s2);
    rot_cols(Q, p, p + 1, 0, nq, c2, s2);
    B(p + 1, p) = 0.0; B(p, p + 1) = 0.0;
    for (int k = 0; k < 2; k++)
        if (B(p + k, p + k) < 0.0) {              // positive diagonal: flip a column (Z diag(+-1))
            for (int i = 0; i < std::min(n, p + 2); i++) A(i, p + k) = -A(i, p + k);
            for (int i = 0; i <= p + k; i++) B(i, p + k) = -B(i, p + k);
            for (int i = 0; i < nq; i++) Z(i, p + k) = -Z(i, p + k);
        }
    return 2;
}

// Eigenvalues (alpha_r + i alpha_i) / beta of a generalized Schur form (S quasi-triangular,
// T upper triangular with standardised 2x2 blocks) -- reference common/math.c:148-176 (dlag2).
// Infinite-eigenvalue chase inside one window (schur/cpu_utils.c:360-425, push_inf_up): B is
// upper triangular with B(from,from) (numerically) zero, A upper Hessenberg.  A column
// rotation moves the zero one position up the diagonal, a row rotation removes the fill it
// leaves below A's sub-diagonal; after from-to steps the zero sits at B(to,to).  With
// deflate != 0 (to is the first row of the active block) a last row rotation annihilates
// A(to+1,to): the pencil splits off the 1x1 block (A(to,to), 0) -- an infinite eigenvalue.
// Q, Z (w x w) accumulate the row / column rotations from the right.  The zero may sit in the
// last row of the window only when that row is the last row of the active block (no fill).
void gep_push_inf_window(int w, double *A_, int lda, double *B_, int ldb, double *Q_, int ldq,
    double *Z_, int ldz, int from, int to, int deflate)
{
    Mat A{A_, lda}, B{B_, ldb}, Q{Q_, ldq}, Z{Z_, ldz};
    for (int i = from; i > to; i--) {
        double const x = B(i - 1, i - 1), y = B(i - 1, i), r = std::hypot(x, y);
        double c = 1.0, s = 0.0;
        if (r != 0.0) { c = y / r; s = -x / r; }         // [x y] G = [0 r]
        rot_cols(A, i - 1, i, 0, std::min(w, i + 2), c, s);
        rot_cols(B, i - 1, i, 0, i - 1, c, s);
        rot_cols(Z, i - 1, i, 0, w, c, s);
        B(i - 1, i) = r; B(i - 1, i - 1) = 0.0;
        if (i + 1 < w) {
            double c2, s2, r2;
            givens(A(i, i - 1), A(i + 1, i - 1), c2, s2, r2);
            rot_rows(A, i, i + 1, i, w, c2, s2);
            rot_rows(B, i, i + 1, i + 1, w, c2, s2);
            rot_cols(Q, i, i + 1, 0, w, c2, s2);
            A(i, i - 1) = r2; A(i + 1, i - 1) = 0.0;
        }
    }
    if (deflate && to + 1 < w) {
        double c2, s2, r2;
        givens(A(to, to), A(to + 1, to), c2, s2, r2);
        rot_rows(A, to, to + 1, to + 1, w, c2, s2);
        rot_rows(B, to, to + 1, to + 1, w, c2, s2);
        rot_cols(Q, to, to + 1, 0, w, c2, s2);
        A(to, to) = r2; A(to + 1, to) = 0.0;
        B(to, to) = 0.0;
    }
}

// The mirror image: the zero at B(from,from) is chased DOWN to the last row of the window
// (LAPACK dhgeqz, "chase the zero to T(ILAST,ILAST)": a row rotation moves the zero one position
// down the diagonal, a column rotation removes the fill below A's sub-diagonal).  from >= 1
// unless row 0 is the first row of the active block.  With deflate != 0 (the window ends the
// active block) a last column rotation annihilates A(w-1,w-2): the 1x1 block (A(w-1,w-1), 0)
// splits off at the bottom.
void gep_push_inf_down_window(int w, double *A_, int lda, double *B_, int ldb, double *Q_, int ldq,
    double *Z_, int ldz, int from, int deflate)
{
    Mat A{A_, lda}, B{B_, ldb}, Q{Q_, ldq}, Z{Z_, ldz};
    double c, s, r;
    for (int jch = from; jch < w - 1; jch++) {
        givens(B(jch, jch + 1), B(jch + 1, jch + 1), c, s, r);
        B(jch, jch + 1) = r; B(jch + 1, jch + 1) = 0.0;
        rot_rows(B, jch, jch + 1, jch + 2, w, c, s);
        rot_rows(A, jch, jch + 1, jch >= 1 ? jch - 1 : 0, w, c, s);
        rot_cols(Q, jch, jch + 1, 0, w, c, s);
        if (jch >= 1) {
            givens(A(jch + 1, jch), A(jch + 1, jch - 1), c, s, r);
            A(jch + 1, jch) = r; A(jch + 1, jch - 1) = 0.0;
            rot_cols(A, jch, jch - 1, 0, jch + 1, c, s);
            rot_cols(B, jch, jch - 1, 0, jch, c, s);
            rot_cols(Z, jch, jch - 1, 0, w, c, s);
        }
    }
    if (deflate && w >= 2) {
        int const il = w - 1;
        givens(A(il, il), A(il, il - 1), c, s, r);
        A(il, il) = r; A(il, il - 1) = 0.0;
        rot_cols(A, il, il - 1, 0, il, c, s);
        rot_cols(B, il, il - 1, 0, il, c, s);
        rot_cols(Z, il, il - 1, 0, w, c, s);
        B(il, il) = 0.0;
    }
}

// The eigenvalues (alpha_r, alpha_i, beta) of a generalized Schur form in diagonal order, as the reference
// returns them (common/tasks.c:1120-1150, schur/cpu_utils.c:3493-3520): (S(i,i), 0, T(i,i)) for a 1 x 1
// block, dlag2's (wr, +-wi, scale) for a 2 x 2 block.
void gep_extract_eigenvalues(int n, const double *S_, int lds, const double *T_, int ldt,
    double *ar, double *ai, double *be)
{
    Mat S{const_cast<double *>(S_), lds}, T{const_cast<double *>(T_), ldt};
    for (int i = 0; i < n; i++) {
        if (i + 1 < n && S(i + 1, i) != 0.0) {
            double s1, s2, w1, w2, wi;
            pencil2_dlag2(&S(i, i), lds, &T(i, i), ldt, s1, s2, w1, w2, wi);
            ar[i] = w1; ai[i] = wi; be[i] = s1;
            ar[i + 1] = w2; ai[i + 1] = -wi; be[i + 1] = s2;
            i++;
        } else { ar[i] = S(i, i); ai[i] = 0.0; be[i] = T(i, i); }
    }
}

// Implicit double-shift QZ on a small Hessenberg-triangular pencil; Q <- Q*U1, Z <- Z*U2 with
// U1^T A U2 quasi-triangular, U1^T B U2 triangular.  Q and Z have nq rows.  Returns 0 or the
// (1-based) row where the iteration limit was hit.
int gep_small_schur(int n, double *A_, int lda, double *B_, int ldb, double *Q_, int ldq,
    double *Z_, int ldz, int nq, double *ar, double *ai, double *be, double thres_b)
{
    Mat A{A_, lda}, B{B_, ldb}, Q{Q_, ldq}, Z{Z_, ldz};
    const double ulp = DBL_EPSILON, safmin = DBL_MIN;
    if (n == 0) return 0;
    // B-side threshold (conf->right_threshold, schur/core.c:2438-2449): an explicit or norm-stable
    // value from the caller, otherwise LAPACK dhgeqz's BTOL for the pencil at hand
    double btol = thres_b;
    if (!(btol > 0.0)) {
        double bn = 0.0;
        for (int j = 0; j < n; j++) for (int i = 0; i <= j; i++) bn = std::hypot(bn, B(i, j));
        btol = std::max(safmin, ulp * bn);
    }
    for (int j = 0; j < n; j++) {               // clean below the (sub)diagonal
        for (int i = j + 2; i < n; i++) A(i, j) = 0.0;
        for (int i = j + 1; i < n; i++) B(i, j) = 0.0;
    }
    int ilast = n - 1, iiter = 0, total = 0;
    const int maxit = 30 * std::max(10, n);
    while (ilast >= 0) {
        // ---- locate the active block [ifirst, ilast]
        int ifirst = 0;
        for (int j = ilast; j >= 1; j--) {
            double tst = std::fabs(A(j, j)) + std::fabs(A(j - 1, j - 1));
            if (tst == 0.0) tst = std::fabs(A(j, j - 1));
            if (std::fabs(A(j, j - 1)) <= std::max(safmin, ulp * tst)) { A(j, j - 1) = 0.0; ifirst = j; break; }
        }
        if (ifirst == ilast) {                  // 1x1 block
            if (B(ilast, ilast) < 0.0) {
                for (int i = 0; i <= ilast; i++) { A(i, ilast) = -A(i, ilast); B(i, ilast) = -B(i, ilast); }
                for (int i = 0; i < nq; i++) Z(i, ilast) = -Z(i, ilast);
            }
            ilast--; iiter = 0; continue;
        }
        // a negligible diagonal entry of B at the bottom of the block: an infinite eigenvalue, split off
        // as LAPACK dhgeqz does (B(ilast,ilast) = 0 exactly, a column rotation annihilates
        // A(ilast,ilast-1)); this is where conf->right_threshold decides
        if (std::fabs(B(ilast, ilast)) < btol) {
            B(ilast, ilast) = 0.0;
            double const x = A(ilast, ilast - 1), y = A(ilast, ilast), r = norm2(x, y);
            double c = 1.0, s = 0.0;
            if (r != 0.0) { c = y / r; s = -x / r; }     // [x y] G = [0 r]
            rot_cols(A, ilast - 1, ilast, 0, ilast + 1, c, s);
            rot_cols(B, ilast - 1, ilast, 0, ilast, c, s);
            rot_cols(Z, ilast - 1, ilast, 0, nq, c, s);
            A(ilast, ilast - 1) = 0.0;
            ilast--; iiter = 0; continue;
        }
        // numerically singular B elsewhere inside the block: perturb to the threshold (see the file header)
        for (int j = ifirst; j < ilast; j++)
            if (std::fabs(B(j, j)) < btol) B(j, j) = (B(j, j) < 0.0) ? -btol : btol;
        if (ifirst == ilast - 1) {              // 2x2 block
            int kind = gep_standardise_2x2(n, A, B, Q, Z, nq, ifirst);
            if (kind == 2) { ilast -= 2; iiter = 0; }
            continue;                            // real pair: two 1x1 blocks on the next passes
        }
        if (++total > maxit) return ilast + 1;
        iiter++;
        // ---- shifts: eigenvalues of the trailing 2x2 pencil (their sum and product)
        int const l = ifirst, m = ilast;
        double sum, prod;
        pencil2_sum_prod(A(m - 1, m - 1), A(m - 1, m), A(m, m - 1), A(m, m),
            B(m - 1, m - 1), B(m - 1, m), B(m, m), sum, prod);
        if (iiter % 10 == 0) {                  // exceptional shift pair
            double e = (std::fabs(A(m, m - 1)) + std::fabs(A(m - 1, m - 2))) / std::fabs(B(m - 1, m - 1));
            sum = 1.5 * e + A(m, m) / B(m, m); prod = 0.4375 * e * e + 0.25 * sum * sum;
        }
        // first column of (A B^-1 - s1)(A B^-1 - s2) (reference create_bulge, cpu_utils.c:880-918)
        double z1_0 = A(l, l) / B(l, l), z1_1 = A(l + 1, l) / B(l, l);
        double t1 = z1_1 / B(l + 1, l + 1), t0 = (z1_0 - B(l, l + 1) * t1) / B(l, l);
        double v[3] = {
            A(l, l) * t0 + A(l, l + 1) * t1 - sum * z1_0 + prod,
            A(l + 1, l) * t0 + A(l + 1, l + 1) * t1 - sum * z1_1,
            A(l + 2, l + 1) * t1 };
        // ---- the QZ sweep (Golub & Van Loan Alg. 7.7.2)
        for (int k = l; k <= m - 2; k++) {
            double hv[3], beta, tau = house_first(3, v, hv, beta);
            int const c0 = std::max(k - 1, l);
            refl_left(A, k, 3, c0, n, hv, tau);
            if (k > l) { A(k + 1, k - 1) = 0.0; A(k + 2, k - 1) = 0.0; }
            refl_left(B, k, 3, k, n, hv, tau);
            refl_right(Q, k, 3, 0, nq, hv, tau);
            // zero B(k+2,k), B(k+2,k+1)
            double row[3] = { B(k + 2, k), B(k + 2, k + 1), B(k + 2, k + 2) }, zv[3];
            double tz = house_last(3, row, zv, beta);
            refl_right(A, k, 3, 0, std::min(k + 4, m + 1), zv, tz);
            refl_right(B, k, 3, 0, k + 3, zv, tz);
            refl_right(Z, k, 3, 0, nq, zv, tz);
            B(k + 2, k) = 0.0; B(k + 2, k + 1) = 0.0;
            // zero B(k+1,k)
            double row2[2] = { B(k + 1, k), B(k + 1, k + 1) }, zw[2];
            double tw = house_last(2, row2, zw, beta);
            refl_right(A, k, 2, 0, std::min(k + 4, m + 1), zw, tw);
            refl_right(B, k, 2, 0, k + 2, zw, tw);
            refl_right(Z, k, 2, 0, nq, zw, tw);
            B(k + 1, k) = 0.0;
            v[0] = A(k + 1, k); v[1] = A(k + 2, k); v[2] = (k < m - 2) ? A(k + 3, k) : 0.0;
        }
        {   // last step: rows m-1, m
            double hv[2], beta, tau = house_first(2, v, hv, beta);
            refl_left(A, m - 1, 2, m - 2, n, hv, tau);
            A(m, m - 2) = 0.0;
            refl_left(B, m - 1, 2, m - 1, n, hv, tau);
            refl_right(Q, m - 1, 2, 0, nq, hv, tau);
            double row2[2] = { B(m, m - 1), B(m, m) }, zw[2];
            double tw = house_last(2, row2, zw, beta);
            refl_right(A, m - 1, 2, 0, m + 1, zw, tw);
            refl_right(B, m - 1, 2, 0, m + 1, zw, tw);
            refl_right(Z, m - 1, 2, 0, nq, zw, tw);
            B(m, m - 1) = 0.0;
        }
    }
    gep_extract_eigenvalues(n, A_, lda, B_, ldb, ar, ai, be);
    return 0;
}

// Reduction of (A,B), B upper triangular, to Hessenberg-triangular form on rows/columns
// [ilo, ihi] by Givens rotations (LAPACK dgghrd, unblocked; reference cpu_utils.c:2681-2684).
// Row rotations act on columns up to n-1, column rotations on rows 0..ihi; Q, Z have nq rows.
void gep_ht_reduce(int n, int ilo, int ihi, double *A_, int lda, double *B_, int ldb,
    double *Q_, int ldq, double *Z_, int ldz, int nq)
{
    Mat A{A_, lda}, B{B_, ldb}, Q{Q_, ldq}, Z{Z_, ldz};
    for (int jc = ilo; jc <= ihi - 2; jc++) {
        for (int jr = ihi; jr >= jc + 2; jr--) {
            double c, s, r;
            givens(A(jr - 1, jc), A(jr, jc), c, s, r);      // rows jr-1, jr kill A(jr, jc)
            A(jr - 1, jc) = r; A(jr, jc) = 0.0;
            rot_rows(A, jr - 1, jr, jc + 1, n, c, s);
            rot_rows(B, jr - 1, jr, jr - 1, n, c, s);
            rot_cols(Q, jr - 1, jr, 0, nq, c, s);
            givens(B(jr, jr), B(jr, jr - 1), c, s, r);      // columns jr, jr-1 kill B(jr, jr-1)
            B(jr, jr) = r; B(jr, jr - 1) = 0.0;
            rot_cols(A, jr, jr - 1, 0, ihi + 1, c, s);
            rot_cols(B, jr, jr - 1, 0, jr, c, s);
            rot_cols(Z, jr, jr - 1, 0, nq, c, s);
        }
    }
}

// ---- reordering (LAPACK dtgex2 / dtgexc, upward direction; reference starneig_move_block,
// schur/cpu_utils.c:3377-3416) --------------------------------------------------------------
namespace {

// Explicit m x m orthogonal factor of the QR factorisation of the m x k matrix X (ld 4, m <= 4,
// k <= 2): X = U [R; 0].  U is returned column-major with ld 4.
static void small_qr(int m, int k, double const *X, double *Uo)
{
    double W[16];
    for (int c = 0; c < k; c++) for (int r = 0; r < m; r++) W[r + 4 * c] = X[r + 4 * c];
    for (int c = 0; c < m; c++) for (int r = 0; r < m; r++) Uo[r + 4 * c] = (r == c) ? 1.0 : 0.0;
    for (int c = 0; c < k && c < m - 1; c++) {
        double v[4], beta, tau = house_first(m - c, &W[c + 4 * c], v, beta);
        if (tau == 0.0) continue;
        for (int cc = c; cc < k; cc++) {            // W <- H W
            double sdot = 0.0;
            for (int r = 0; r < m - c; r++) sdot += v[r] * W[c + r + 4 * cc];
            sdot *= tau;
            for (int r = 0; r < m - c; r++) W[c + r + 4 * cc] -= sdot * v[r];
        }
        for (int r = 0; r < m; r++) {               // U <- U H
            double sdot = 0.0;
            for (int q = 0; q < m - c; q++) sdot += Uo[r + 4 * (c + q)] * v[q];
            sdot *= tau;
            for (int q = 0; q < m - c; q++) Uo[r + 4 * (c + q)] -= sdot * v[q];
        }
    }
}

// Swaps the adjacent diagonal blocks (n1 x n1 at j, n2 x n2 at j+n1) of the generalized
// Schur form (A,B) by an orthogonal equivalence; Q, Z (nq rows) are updated.  The deflating
// subspaces of the second block come from the generalized Sylvester equation
//   A11 R - L A22 = -A12,  B11 R - L B22 = -B12   (right: [R; I], left: [L; I]),
// solved as a dense system of 2 n1 n2 <= 8 unknowns with complete pivoting.  Returns false
// (nothing changed) when the blocks are too close to swap stably (LAPACK's weak stability test).
// M(0:rows, j:j+m) <- M(0:rows, j:j+m) W, W m x m (m <= 4) with leading dimension 4: the columns are
// distinct and the rows contiguous, so the row loop vectorises
template <int MM>
static void right_apply_m(Mat M, int rows, int j, const double *W)
{
    double *__restrict__ x0 = &M(0, j), *__restrict__ x1 = &M(0, j + 1);
    double *__restrict__ x2 = MM > 2 ? &M(0, j + 2) : nullptr, *__restrict__ x3 = MM > 3 ? &M(0, j + 3) : nullptr;
    for (int r = 0; r < rows; r++) {
        double const a0 = x0[r], a1 = x1[r], a2 = MM > 2 ? x2[r] : 0.0, a3 = MM > 3 ? x3[r] : 0.0;
        double o[4];
        #pragma unroll
        for (int c = 0; c < MM; c++) {
            double v = a0 * W[0 + 4 * c] + a1 * W[1 + 4 * c];
            if (MM > 2) v += a2 * W[2 + 4 * c];
            if (MM > 3) v += a3 * W[3 + 4 * c];
            o[c] = v;
        }
        x0[r] = o[0]; x1[r] = o[1];
        if (MM > 2) x2[r] = o[2];
        if (MM > 3) x3[r] = o[3];
    }
}
static void right_apply(Mat M, int rows, int j, int m, const double *W)
{
    if (m == 2) right_apply_m<2>(M, rows, j, W);
    else if (m == 3) right_apply_m<3>(M, rows, j, W);
    else right_apply_m<4>(M, rows, j, W);
}

static bool gep_swap_adjacent(int nw, Mat A, Mat B, Mat Q, Mat Z, int nq, int j, int n1, int n2)
{
    int const m = n1 + n2, nn = n1 * n2, N = 2 * nn;
    double S[16], T[16];
    // Frobenius norm of the local pencil: a plain sum of squares unless an entry is near the ends of the
    // exponent range (32 hypot calls cost more than the rest of a 1x1-1x1 swap)
    double fro = 0.0, big = 0.0;
    for (int c = 0; c < m; c++)
        for (int r = 0; r < m; r++) {
            double const a = A(j + r, j + c), b = B(j + r, j + c);
            S[r + 4 * c] = a; T[r + 4 * c] = b;
            fro += a * a + b * b; big = std::max(big, std::max(std::fabs(a), std::fabs(b)));
        }
    bool const plain = big < 1e140 && big > 1e-140;
    if (plain) fro = std::sqrt(fro);
    else {
        fro = 0.0;
        for (int c = 0; c < m; c++) for (int r = 0; r < m; r++) fro = std::hypot(fro, std::hypot(S[r + 4 * c], T[r + 4 * c]));
    }
    double K[64], u[8];
    for (int i = 0; i < 64; i++) K[i] = 0.0;
    for (int k = 0; k < n2; k++)
        for (int i = 0; i < n1; i++) {
            int const e = i + n1 * k;
            for (int p = 0; p < n1; p++) {
                K[e + N * (p + n1 * k)] += S[i + 4 * p];
                K[nn + e + N * (p + n1 * k)] += T[i + 4 * p];
            }
            for (int q = 0; q < n2; q++) {
                K[e + N * (nn + i + n1 * q)] -= S[(n1 + q) + 4 * (n1 + k)];
                K[nn + e + N * (nn + i + n1 * q)] -= T[(n1 + q) + 4 * (n1 + k)];
            }
            u[e] = -S[i + 4 * (n1 + k)];
            u[nn + e] = -T[i + 4 * (n1 + k)];
        }
    // Gaussian elimination with complete pivoting
    int perm[8];
    for (int i = 0; i < N; i++) perm[i] = i;
    double kmax = 0.0;
    for (int i = 0; i < N * N; i++) kmax = std::max(kmax, std::fabs(K[i]));
    if (kmax == 0.0) return false;
    for (int c = 0; c < N; c++) {
        int pr = c, pc = c; double best = -1.0;
        for (int cc = c; cc < N; cc++)
            for (int r = c; r < N; r++)
                if (std::fabs(K[r + N * cc]) > best) { best = std::fabs(K[r + N * cc]); pr = r; pc = cc; }
        if (best <= 8.0 * DBL_EPSILON * kmax) return false;     // (nearly) common eigenvalue
        if (pr != c) { for (int cc = 0; cc < N; cc++) std::swap(K[c + N * cc], K[pr + N * cc]); std::swap(u[c], u[pr]); }
        if (pc != c) { for (int r = 0; r < N; r++) std::swap(K[r + N * c], K[r + N * pc]); std::swap(perm[c], perm[pc]); }
        for (int r = c + 1; r < N; r++) {
            double const f = K[r + N * c] / K[c + N * c];
            if (f == 0.0) continue;
            for (int cc = c; cc < N; cc++) K[r + N * cc] -= f * K[c + N * cc];
            u[r] -= f * u[c];
        }
    }
    double sol[8];
    for (int c = N - 1; c >= 0; c--) {
        double v = u[c];
        for (int cc = c + 1; cc < N; cc++) v -= K[c + N * cc] * sol[perm[cc]];
        sol[perm[c]] = v / K[c + N * c];
    }
    // X = [R; I], Y = [L; I]  (m x n2)
    double X[16], Y[16], Zl[16], Ql[16];
    for (int k = 0; k < n2; k++) {
        for (int i = 0; i < n1; i++) { X[i + 4 * k] = sol[i + n1 * k]; Y[i + 4 * k] = sol[nn + i + n1 * k]; }
        for (int q = 0; q < n2; q++) { X[n1 + q + 4 * k] = (q == k) ? 1.0 : 0.0; Y[n1 + q + 4 * k] = (q == k) ? 1.0 : 0.0; }
    }
    small_qr(m, n2, X, Zl);
    small_qr(m, n2, Y, Ql);
    // weak stability test on the local pencil: (2,1) blocks of Ql^T S Zl and Ql^T T Zl
    double SZ[16], TZ[16], S2[16], T2[16];
    for (int c = 0; c < m; c++)
        for (int r = 0; r < m; r++) {
            double a = 0.0, b = 0.0;
            for (int p = 0; p < m; p++) { a += S[r + 4 * p] * Zl[p + 4 * c]; b += T[r + 4 * p] * Zl[p + 4 * c]; }
            SZ[r + 4 * c] = a; TZ[r + 4 * c] = b;
        }
    double low = 0.0;
    for (int c = 0; c < m; c++)
        for (int r = 0; r < m; r++) {
            double a = 0.0, b = 0.0;
            for (int p = 0; p < m; p++) { a += Ql[p + 4 * r] * SZ[p + 4 * c]; b += Ql[p + 4 * r] * TZ[p + 4 * c]; }
            S2[r + 4 * c] = a; T2[r + 4 * c] = b;
            if (r >= n2 && c < n2) { if (plain) low += a * a + b * b; else low = std::hypot(low, std::hypot(a, b)); }
        }
    if (plain) low = std::sqrt(low);
    if (low > std::max(20.0 * DBL_EPSILON * fro, DBL_MIN)) return false;
    // ---- apply to the window: rows j..j+m-1 <- Ql^T ., columns j..j+m-1 <- . Zl
    double tmp[4];
    for (int c = j; c < nw; c++) {
        for (int r = 0; r < m; r++) { double a = 0.0; for (int p = 0; p < m; p++) a += Ql[p + 4 * r] * A(j + p, c); tmp[r] = a; }
        for (int r = 0; r < m; r++) A(j + r, c) = tmp[r];
        for (int r = 0; r < m; r++) { double a = 0.0; for (int p = 0; p < m; p++) a += Ql[p + 4 * r] * B(j + p, c); tmp[r] = a; }
        for (int r = 0; r < m; r++) B(j + r, c) = tmp[r];
    }
    right_apply(A, j + m, j, m, Zl); right_apply(B, j + m, j, m, Zl); right_apply(Z, nq, j, m, Zl);
    right_apply(Q, nq, j, m, Ql);
    for (int c = 0; c < n2; c++) for (int r = n2; r < m; r++) { A(j + r, j + c) = 0.0; B(j + r, j + c) = 0.0; }
    // ---- restore the standard form of the two diagonal blocks
    auto fix_block = [&](int p, int bs) {
        if (bs == 2) {
            double c, sn, r;
            givens(B(p, p), B(p + 1, p), c, sn, r);          // B block upper triangular again
            rot_rows(A, p, p + 1, p, nw, c, sn);
            rot_rows(B, p, p + 1, p, nw, c, sn);
            rot_cols(Q, p, p + 1, 0, nq, c, sn);
            B(p + 1, p) = 0.0;
            if (gep_standardise_2x2(nw, A, B, Q, Z, nq, p) == 2) return;
            bs = 1;                                          // split into two real eigenvalues
            if (B(p + 1, p + 1) < 0.0) {
                for (int i = 0; i <= p + 1; i++) { A(i, p + 1) = -A(i, p + 1); B(i, p + 1) = -B(i, p + 1); }
                for (int i = 0; i < nq; i++) Z(i, p + 1) = -Z(i, p + 1);
            }
        }
        if (B(p, p) < 0.0) {
            for (int i = 0; i <= p; i++) { A(i, p) = -A(i, p); B(i, p) = -B(i, p); }
            for (int i = 0; i < nq; i++) Z(i, p) = -Z(i, p);
        }
    };
    fix_block(j, n2);
    fix_block(j + n2, n1);
    return true;
}

} // namespace

// Moves the diagonal block that starts at row `from` up to row `to` by adjacent swaps.
// Returns the row where it ended up (== to unless a swap was rejected).
int gep_move_block_up(int nw, double *A_, int lda, double *B_, int ldb, double *Q_, int ldq,
    double *Z_, int ldz, int nq, int from, int to)
{
    Mat A{A_, lda}, B{B_, ldb}, Q{Q_, ldq}, Z{Z_, ldz};
    int here = from;
    int nbf = (here + 1 < nw && A(here + 1, here) != 0.0) ? 2 : 1;
    while (here > to) {
        int nbabove = (here - 2 >= 0 && A(here - 1, here - 2) != 0.0) ? 2 : 1;
        if (here - nbabove < to) break;
        int j1 = here - nbabove;
        if (!gep_swap_adjacent(nw, A, B, Q, Z, nq, j1, nbabove, nbf)) break;
        here = j1;
        if (nbf == 2 && A(here + 1, here) == 0.0) {
            // the moving 2x2 block split into two 1x1 blocks: move them one at a time
            int a = gep_move_block_up(nw, A_, lda, B_, ldb, Q_, ldq, Z_, ldz, nq, here, to);
            if (a != to) return a;
            gep_move_block_up(nw, A_, lda, B_, ldb, Q_, ldq, Z_, ldz, nq, here + 1, to + 1);
            return a;
        }
    }
    return here;
}


// Shifts of the next sweep: the finite, non-zero eigenvalues of the leading hi x hi part of a window in
// generalized Schur form, ordered by magnitude, complex pairs adjacent (schur/cpu_utils.c:3493-3594).
int gep_window_shifts(int hi, double const *A_, int lda, double const *B_, int ldb, double *sr, double *si)
{
    std::vector<double> ar(hi), ai(hi), be(hi);
    gep_extract_eigenvalues(hi, A_, lda, B_, ldb, ar.data(), ai.data(), be.data());
    std::vector<double> wr, wi;
    for (int k = 0; k < hi; k++)
        if (be[k] != 0.0) {
            double re = ar[k] / be[k], im = ai[k] / be[k];
            if (std::isfinite(re) && std::isfinite(im) && !(re == 0.0 && im == 0.0)) { wr.push_back(re); wi.push_back(im); }
        }
    int const cnt = (int)wr.size();
    std::vector<int> idx(cnt);
    for (int k = 0; k < cnt; k++) idx[k] = k;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
        return std::fabs(wr[a]) + std::fabs(wi[a]) < std::fabs(wr[b]) + std::fabs(wi[b]); });
    for (int k = 0; k < cnt; k++) { sr[k] = wr[idx[k]]; si[k] = wi[idx[k]]; }
    for (int k = 0; k + 2 < cnt; k += 2)
        if (si[k] != -si[k + 1]) {
            double r0 = sr[k], i0 = si[k];
            sr[k] = sr[k + 1]; sr[k + 1] = sr[k + 2]; sr[k + 2] = r0;
            si[k] = si[k + 1]; si[k + 1] = si[k + 2]; si[k + 2] = i0;
        }
    return cnt;
}

// ---- one deflation window of the blocked AED on a pencil (the generalized twin of host::deflate_window;
// reference schur/cpu.c:638-1006 starneig_cpu_deflate with B != NULL, driven by schur/core.c:1070-1252) ----
// (A, B): w x w diagonal window of the AED window's generalized Schur form; the bottom `carried` rows hold
// blocks an earlier deflation window found undeflatable, the rows above them are unchecked.  spike[0:w] is
// the window's segment of the spike sub * Q_aed(0,:).  Q, Z: identity on entry, the accumulated swaps on
// return.  The carried blocks go to the top of the window, then the unchecked blocks are tested from the
// bottom (a block is tested when it is the last undeflated one; an undeflatable block joins the carried
// ones at the top).  On return [0, *undeflated) holds the undeflatable blocks, [*undeflated, w) the deflated
// ones, spike <- spike * Q.  Returns 0, or 1 if a swap was rejected (everything above counts as undeflatable).
int gep_deflate_window(int w, double *A_, int lda, double *B_, int ldb, double *Q_, int ldq, double *Z_, int ldz,
    double *spike, double sub, double thres, int carried, int *undeflated)
{
    Mat A{A_, lda}, Q{Q_, ldq};
    int rc = 0, top = 0;
    bool tested = false;
    for (int i = w - carried; i < w;) {
        int const bs = (i + 1 < w && A(i + 1, i) != 0.0) ? 2 : 1;
        if (i > top) {
            int const at = gep_move_block_up(w, A_, lda, B_, ldb, Q_, ldq, Z_, ldz, w, i, top);
            if (at != top) { *undeflated = w; rc = 1; tested = true; break; }   // nothing can be tested behind it
        }
        top += bs; i += bs;
    }
    if (!tested) {
        const double ulp = DBL_EPSILON, smlnum = DBL_MIN * ((double)w / ulp);
        auto cur = [&](int col) { double v = 0.0; for (int k = 0; k < w; k++) v += spike[k] * Q(k, col); return v; };
        int i = w - 1;
        while (top <= i) {
            bool const two = (top <= i - 1 && A(i, i - 1) != 0.0);
            double sp = std::fabs(cur(i));
            if (two) sp = std::max(sp, std::fabs(cur(i - 1)));
            bool deflatable;
            if (thres > 0.0) deflatable = sp < thres;
            else {
                double foo = std::fabs(A(i, i));
                if (two) foo += std::sqrt(std::fabs(A(i, i - 1))) * std::sqrt(std::fabs(A(i - 1, i)));
                if (foo == 0.0) foo = std::fabs(sub);
                deflatable = sp < std::max(smlnum, ulp * foo);
            }
            int const bs = two ? 2 : 1;
            if (deflatable) i -= bs;
            else {
                int const at = gep_move_block_up(w, A_, lda, B_, ldb, Q_, ldq, Z_, ldz, w, i - bs + 1, top);
                if (at != top) { top = i + 1; rc = 1; break; }
                top += bs;
            }
        }
        *undeflated = top;
    }
    std::vector<double> ns(w);
    for (int j = 0; j < w; j++) { double v = 0.0; for (int k = 0; k < w; k++) v += spike[k] * Q(k, j); ns[j] = v; }
    for (int j = 0; j < w; j++) spike[j] = ns[j];
    return rc;
}

// The marked diagonal blocks of a w x w window of a generalized Schur form to the top of the window, in their
// order (the generalized twin of host::reorder_window; LAPACK dtgsen's loop of dtgexc calls).  sel[i] != 0
// marks the rows of selected blocks; on return sel holds the marks of the rows in their new order.  Returns
// the number of rows of selected blocks now at the top; *failed is set when a swap was rejected.
int gep_reorder_window(int w, double *A_, int lda, double *B_, int ldb, double *Q_, int ldq, double *Z_, int ldz,
    int *sel, int *failed)
{
    Mat A{A_, lda};
    *failed = 0;
    int top = 0, i = 0;
    while (i < w) {
        int const bs = (i + 1 < w && A(i + 1, i) != 0.0) ? 2 : 1;
        bool const marked = sel[i] != 0 || (bs == 2 && sel[i + 1] != 0);
        if (!marked) { i += bs; continue; }
        if (i > top) {
            int const at = gep_move_block_up(w, A_, lda, B_, ldb, Q_, ldq, Z_, ldz, w, i, top);
            for (int r = i + bs - 1; r >= at + bs; r--) sel[r] = 0;
            for (int r = at; r < at + bs; r++) sel[r] = 1;
            if (at != top) { *failed = 1; return top; }
        } else for (int r = i; r < i + bs; r++) sel[r] = 1;
        top += bs;
        i += bs;
    }
    return top;
}

// Aggressive early deflation on a host window of the pencil (reference
// perform_aggressively_deflate, cpu_utils.c:2837-3046, generalized branches): eigenvalues
// whose spike entries are below the threshold are deflated, the others are reordered to the
// top of the window.  On return (S,T) = [HT (ns x ns) | *; 0 | Schur (nd x nd)], Q/Z the accumulated
// transformations, spike[0] the new coupling entry, shifts as complex numbers alpha/beta.
AedResult gep_aed_window(int nw, double *A_, int lda, double *B_, int ldb, double *Q_, int ldq,
    double *Z_, int ldz, double sub, double thres, double *spike, double *sr, double *si, double thres_b)
{
    AedResult res{0, 0, 0};
    Mat A{A_, lda}, B{B_, ldb}, Q{Q_, ldq}, Z{Z_, ldz};
    for (int j = 0; j < nw; j++) for (int i = 0; i < nw; i++) { Q(i, j) = (i == j); Z(i, j) = (i == j); }
    std::vector<double> ar(nw), ai(nw), be(nw);
    bool const prof = tuning().aed_profile;
    static double t_schur = 0, t_reorder = 0, t_rest = 0; static int calls = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double const t0 = prof ? now() : 0.0;
    int info = gep_small_schur(nw, A_, lda, B_, ldb, Q_, ldq, Z_, ldz, nw, ar.data(), ai.data(), be.data(), thres_b);
    if (info != 0) { res.failed = 1; return res; }
    double const t1 = prof ? now() : 0.0;
    // deflation scan from the bottom; undeflatable blocks are moved to the top of the window
    // (they accumulate in [0, top)) so that every converged eigenvalue can be deflated
    int top = 0, i = nw - 1;
    while (top <= i) {
        bool two = (top <= i - 1 && A(i, i - 1) != 0.0);
        double sp = std::fabs(sub * Q(0, i));
        if (two) sp = std::max(sp, std::fabs(sub * Q(0, i - 1)));
        bool ok;
        if (thres > 0.0)        // norm-stable criterion (cpu_utils.c:2891-2931)
            ok = sp < thres;
        else {                  // LAPACK-style criterion (:2937-2988)
            double const ulp = DBL_EPSILON, smlnum = DBL_MIN * ((double)nw / ulp);
            double foo = std::fabs(A(i, i));
            if (two) foo += std::sqrt(std::fabs(A(i, i - 1))) * std::sqrt(std::fabs(A(i - 1, i)));
            if (foo == 0.0) foo = std::fabs(sub);
            ok = sp < std::max(smlnum, ulp * foo);
        }
        int const bs = two ? 2 : 1;
        if (ok) i -= bs;
        else {
            int const from = i - bs + 1;
            int at = gep_move_block_up(nw, A_, lda, B_, ldb, Q_, ldq, Z_, ldz, nw, from, top);
            if (at != top) { top = i + 1; break; }   // swap rejected: nothing above `i` is examined
            top += bs;
        }
    }
    int const ns = top, nd = nw - top;
    res.deflated = nd;
    double const t2 = prof ? now() : 0.0;
    struct Report { bool on; double a, b, c; decltype(now) &clk; int nw, nd; ~Report() {
        if (!on) return;
        t_schur += b - a; t_reorder += c - b; t_rest += clk() - c; calls++;
        if (calls % 50 == 0) {
            fprintf(stderr, "[gep aed] calls %d, the last 50 per call: QZ %.3f ms, deflation %.3f ms, shifts + restoration %.3f ms (nw %d, last deflated %d)\n",
                calls, 20.0 * t_schur, 20.0 * t_reorder, 20.0 * t_rest, nw, nd);
            t_schur = t_reorder = t_rest = 0;
        } } } report{prof, t0, t1, t2, now, nw, nd};
    for (int j = 0; j < nw; j++) spike[j] = sub * Q(0, j);
    // shifts: finite eigenvalues of the undeflated part (of everything if that is too small)
    res.shifts = gep_window_shifts((ns >= 2) ? ns : nw, A_, lda, B_, ldb, sr, si);
    if (nd == 0) return res;
    for (int j = ns; j < nw; j++) spike[j] = 0.0;
    if (ns > 1 && sub != 0.0) {
        // padded pencil: row/column 0 carries the spike, then dgghrd on [0, ns]
        int const np = nw + 1;
        std::vector<double> Ap((size_t)np * np, 0.0), Bp((size_t)np * np, 0.0), Qp((size_t)np * np, 0.0), Zp((size_t)np * np, 0.0);
        Mat PA{Ap.data(), np}, PB{Bp.data(), np}, PQ{Qp.data(), np}, PZ{Zp.data(), np};
        PB(0, 0) = 1.0; PQ(0, 0) = 1.0; PZ(0, 0) = 1.0;
        for (int j = 0; j < nw; j++)
            for (int r = 0; r < nw; r++) {
                PA(r + 1, j + 1) = A(r, j); PB(r + 1, j + 1) = B(r, j);
                PQ(r + 1, j + 1) = Q(r, j); PZ(r + 1, j + 1) = Z(r, j);
            }
        for (int r = 0; r < ns; r++) PA(r + 1, 0) = spike[r];
        gep_ht_reduce(np, 0, ns, Ap.data(), np, Bp.data(), np, Qp.data(), np, Zp.data(), np, np);
        for (int j = 0; j < nw; j++)
            for (int r = 0; r < nw; r++) {
                A(r, j) = PA(r + 1, j + 1); B(r, j) = PB(r + 1, j + 1);
                Q(r, j) = PQ(r + 1, j + 1); Z(r, j) = PZ(r + 1, j + 1);
            }
        spike[0] = PA(1, 0);
        for (int r = 1; r < ns; r++) spike[r] = 0.0;
        // exact structure
        for (int j = 0; j < ns; j++) {
            for (int r = j + 2; r < nw; r++) A(r, j) = 0.0;
            for (int r = j + 1; r < nw; r++) B(r, j) = 0.0;
        }
    }
    return res;
}

}} // namespace sn::host

#ifdef SN_TEST_HOOKS   // compiled into libstarneig_amd_test.so only (csrc/Makefile), never into the product library
// ---- test hooks (host-only; NOT part of the public C-ABI, used by tests/ on CPU) -------
extern "C" {
__attribute__((visibility("default")))
int sn_internal_gep_deflate_window(int w, double *A, int lda, double *B, int ldb, double *Q, int ldq, double *Z, int ldz,
    double *spike, double sub, double thres, int carried, int *undeflated)
{ return sn::host::gep_deflate_window(w, A, lda, B, ldb, Q, ldq, Z, ldz, spike, sub, thres, carried, undeflated); }
__attribute__((visibility("default")))
int sn_internal_gep_reorder_window(int w, double *A, int lda, double *B, int ldb, double *Q, int ldq, double *Z, int ldz,
    int *sel, int *failed)
{ return sn::host::gep_reorder_window(w, A, lda, B, ldb, Q, ldq, Z, ldz, sel, failed); }
__attribute__((visibility("default")))
void sn_internal_gep_extract_eigenvalues(int n, double const *S, int lds, double const *T, int ldt,
    double *ar, double *ai, double *be)
{ sn::host::gep_extract_eigenvalues(n, S, lds, T, ldt, ar, ai, be); }
__attribute__((visibility("default")))
void sn_internal_gep_push_inf_down_window(int w, double *A, int lda, double *B, int ldb, double *Q, int ldq,
    double *Z, int ldz, int from, int deflate)
{ sn::host::gep_push_inf_down_window(w, A, lda, B, ldb, Q, ldq, Z, ldz, from, deflate); }
__attribute__((visibility("default")))
void sn_internal_gep_push_inf_window(int w, double *A, int lda, double *B, int ldb, double *Q, int ldq,
    double *Z, int ldz, int from, int to, int deflate)
{ sn::host::gep_push_inf_window(w, A, lda, B, ldb, Q, ldq, Z, ldz, from, to, deflate); }
__attribute__((visibility("default")))
int sn_internal_gep_small_schur(int n, double *A, int lda, double *B, int ldb, double *Q, int ldq,
    double *Z, int ldz, double *ar, double *ai, double *be)
{ return sn::host::gep_small_schur(n, A, lda, B, ldb, Q, ldq, Z, ldz, n, ar, ai, be); }
__attribute__((visibility("default")))
void sn_internal_gep_ht_reduce(int n, int ilo, int ihi, double *A, int lda, double *B, int ldb,
    double *Q, int ldq, double *Z, int ldz)
{ sn::host::gep_ht_reduce(n, ilo, ihi, A, lda, B, ldb, Q, ldq, Z, ldz, n); }
__attribute__((visibility("default")))
int sn_internal_gep_move_block_up(int n, double *A, int lda, double *B, int ldb, double *Q, int ldq,
    double *Z, int ldz, int from, int to)
{ return sn::host::gep_move_block_up(n, A, lda, B, ldb, Q, ldq, Z, ldz, n, from, to); }
__attribute__((visibility("default")))
int sn_internal_gep_aed_window(int nw, double *A, int lda, double *B, int ldb, double *Q, int ldq,
    double *Z, int ldz, double sub, double thres, double *spike, double *sr, double *si, int *out3)
{
    sn::host::AedResult r = sn::host::gep_aed_window(nw, A, lda, B, ldb, Q, ldq, Z, ldz, sub, thres, spike, sr, si);
    out3[0] = r.deflated; out3[1] = r.shifts; out3[2] = r.failed;
    return 0;
}
}
#endif  // SN_TEST_HOOKS
