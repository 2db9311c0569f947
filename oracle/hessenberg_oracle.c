/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement of the reference's blocked Householder Hessenberg reduction
 * (NLAFET/StarNEig v0.2.0-beta.1), following
 *   - the panel / column / update order of src/hessenberg/core.c:399-596
 *     (insert_tasks) and :301-349 (the delayed updates), and
 *   - the arithmetic of the codelets in src/hessenberg/cpu.c:50-560
 *     (prepare_column, compute_column, finish_column, update_trail_right,
 *      update_left_a/_b, update_right_a/_b),
 * with the BLAS/LAPACK calls those codelets make (cblas_dgemv/dtrmv/dgemm/
 * dtrmm/daxpy/dscal, LAPACK dlarfg -- third-party, unpinned in the reference)
 * written out as plain loops.  dlarfg follows the published LAPACK 3.x
 * algorithm (beta = -sign(alpha)*||x||, safmin rescaling loop).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object.
 *
 * Parity pin: there are no stored golden outputs in the reference for this
 * path.  The oracle is pinned on (1) the reference's own
 * invariant checks (exact zeros below the sub-diagonal, test/common/hooks.c:
 * 434-456; residual and orthogonality in units of u, test/common/checks.c:
 * 180-208) and (2) LAPACK dgehrd+dorghr -- the comparator the reference test
 * driver itself offers (test/hessenberg/solvers.c:231-283) -- through fixtures
 * in tests/golden generated with scipy.  Bitwise parity is unpinned (and is
 * not reproducible by the reference itself: commutative accumulation,
 * hessenberg/tasks.c:374,515,622).
 *
 * All matrices column-major.  Optional OpenMP only splits independent output
 * columns/rows; it never changes a summation order.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define MIN(a,b) ((a) < (b) ? (a) : (b))
#define MAX(a,b) ((a) > (b) ? (a) : (b))

/* test/common/common.c:56-59 */
static unsigned long lcg_state = 2019;
void oracle_init_prand(unsigned int seed) { lcg_state = seed; }
int oracle_prand(void)
{
    return (int)(lcg_state = ((lcg_state * 1103515245UL) + 12345UL) & 0x7fffffffUL);
}
#define PRAND_MAX 0x7fffffff

/* test/common/init.c:108-120 (fullpos), :93-106 (full), :159-175 (hessenberg) */
void oracle_fill_random_fullpos(int m, int n, double *A, int ld)
{
    for (int j = 0; j < n; j++)
        for (int i = 0; i < m; i++)
            A[(size_t)j*ld+i] = 1.0*oracle_prand()/PRAND_MAX;
}
void oracle_fill_random_full(int m, int n, double *A, int ld)
{
    for (int j = 0; j < n; j++)
        for (int i = 0; i < m; i++)
            A[(size_t)j*ld+i] = 2.0*(1.0*oracle_prand()/PRAND_MAX)-1.0;
}
void oracle_fill_random_hessenberg(int n, double *A, int ld)
{
    for (int j = 0; j < n; j++) {
        int end = MIN(n, j+2);
        for (int i = 0; i < end; i++)
            A[(size_t)j*ld+i] = 2.0*(1.0*oracle_prand()/PRAND_MAX)-1.0;
        for (int i = end; i < n; i++)
            A[(size_t)j*ld+i] = 0.0;
    }
}

/* LAPACK dnrm2 (scaled sum of squares form) */
static double nrm2(int n, double const *x)
{
    double scale = 0.0, ssq = 1.0;
    for (int i = 0; i < n; i++) {
        if (x[i] != 0.0) {
            double a = fabs(x[i]);
            if (scale < a) { ssq = 1.0 + ssq*(scale/a)*(scale/a); scale = a; }
            else           { ssq += (a/scale)*(a/scale); }
        }
    }
    return scale*sqrt(ssq);
}

static double lapy2(double x, double y)
{
    double xa = fabs(x), ya = fabs(y);
    double w = MAX(xa, ya), z = MIN(xa, ya);
    if (z == 0.0) return w;
    return w*sqrt(1.0+(z/w)*(z/w));
}

/* LAPACK dlarfg: on exit *alpha = beta, x = v(2:n), returns tau. */
double oracle_dlarfg(int n, double *alpha, double *x)
{
    if (n <= 1) return 0.0;
    double xnorm = nrm2(n-1, x);
    if (xnorm == 0.0) return 0.0;
    double beta = -copysign(lapy2(*alpha, xnorm), *alpha);
    double const safmin = DBL_MIN / (DBL_EPSILON*0.5);
    double const rsafmn = 1.0/safmin;
    int knt = 0;
    if (fabs(beta) < safmin) {
        do {
            knt++;
            for (int i = 0; i < n-1; i++) x[i] *= rsafmn;
            beta *= rsafmn;
            *alpha *= rsafmn;
        } while (fabs(beta) < safmin && knt < 20);
        xnorm = nrm2(n-1, x);
        beta = -copysign(lapy2(*alpha, xnorm), *alpha);
    }
    double tau = (beta-*alpha)/beta;
    double s = 1.0/(*alpha-beta);
    for (int i = 0; i < n-1; i++) x[i] *= s;
    for (int j = 0; j < knt; j++) beta *= safmin;
    *alpha = beta;
    return tau;
}

/* C(m x n) -= A(m x k) * B(n x k)^T   -- cblas_dgemm(N,T) of cpu.c:315,433,552 */
static void gemm_nt_sub(int m, int n, int k,
    double const *A, int lda, double const *B, int ldb, double *C, int ldc)
{
    #pragma omp parallel for schedule(static)
    for (int j = 0; j < n; j++) {
        double *c = C+(size_t)j*ldc;
        for (int l = 0; l < k; l++) {
            double b = B[(size_t)l*ldb+j];
            double const *a = A+(size_t)l*lda;
            for (int i = 0; i < m; i++) c[i] -= a[i]*b;
        }
    }
}

/* W(n x k) = A(m x n)^T * V(m x k)   -- cblas_dgemm(T,N) of cpu.c:373 */
static void gemm_tn(int m, int n, int k,
    double const *A, int lda, double const *V, int ldv, double *W, int ldw)
{
    #pragma omp parallel for schedule(static)
    for (int j = 0; j < n; j++) {
        double const *a = A+(size_t)j*lda;
        for (int l = 0; l < k; l++) {
            double const *v = V+(size_t)l*ldv;
            double s = 0.0;
            for (int i = 0; i < m; i++) s += a[i]*v[i];
            W[(size_t)l*ldw+j] = s;
        }
    }
}

/* W(m x k) = A(m x n) * V(n x k)   -- cblas_dgemm(N,N) of cpu.c:492 */
static void gemm_nn(int m, int n, int k,
    double const *A, int lda, double const *V, int ldv, double *W, int ldw)
{
    #pragma omp parallel for schedule(static)
    for (int l = 0; l < k; l++) {
        double *w = W+(size_t)l*ldw;
        for (int i = 0; i < m; i++) w[i] = 0.0;
        for (int j = 0; j < n; j++) {
            double v = V[(size_t)l*ldv+j];
            double const *a = A+(size_t)j*lda;
            for (int i = 0; i < m; i++) w[i] += a[i]*v;
        }
    }
}

/* W(m x k) <- W * T, T upper triangular k x k  -- cblas_dtrmm(R,U,N,NonUnit) */
static void trmm_right_upper(int m, int k, double const *T, int ldt,
    double *W, int ldw)
{
    for (int j = k-1; j >= 0; j--) {
        double *wj = W+(size_t)j*ldw;
        double t = T[(size_t)j*ldt+j];
        for (int i = 0; i < m; i++) wj[i] *= t;
        for (int l = 0; l < j; l++) {
            double tl = T[(size_t)j*ldt+l];
            double const *wl = W+(size_t)l*ldw;
            for (int i = 0; i < m; i++) wj[i] += wl[i]*tl;
        }
    }
}

struct panel {
    int i, nb, m;
    double *P, *V, *T;
    struct panel *next;
};

/*
 * Reduces columns [begin,end) of the n x n matrix A to Hessenberg form and
 * accumulates Q <- Q*U.  Returns 0, or -1 on allocation failure.
 */
int oracle_hessenberg(int n, int begin, int end, int panel_width,
    double *A, int ldA, double *Q, int ldQ)
{
    struct panel *head = NULL, *tail = NULL;

    for (int i = begin; i < end-1; i += panel_width) {          /* core.c:399 */
        int const nb = MIN(panel_width, end-i-1);               /* core.c:400 */
        int const m = end-i-1;
        size_t const ld = m;
        double *P = malloc(ld*nb*sizeof(double));
        double *V = calloc(ld*nb, sizeof(double));              /* core.c:455 */
        double *Y = malloc(ld*nb*sizeof(double));
        double *T = calloc((size_t)nb*nb, sizeof(double));
        double *y = malloc(ld*sizeof(double));
        if (!P || !V || !Y || !T || !y) return -1;

        for (int j = 0; j < nb; j++)                            /* core.c:451 */
            memcpy(P+j*ld, A+(size_t)(i+j)*ldA+i+1, m*sizeof(double));

        for (int j = 0; j < nb; j++) {                          /* core.c:461 */
            double *p = P+j*ld;
            /* ---- prepare_column, cpu.c:95-131 ---- */
            if (0 < j) {
                /* p -= Y(:,0:j) * V(j-1,0:j)^T          cpu.c:98-99 */
                for (int l = 0; l < j; l++) {
                    double s = V[l*ld+j-1];
                    double const *yl = Y+l*ld;
                    for (int r = 0; r < m; r++) p[r] -= yl[r]*s;
                }
                double *w = T+(size_t)(nb-1)*nb;   /* workspace, cpu.c:106 */
                /* w = V1^T p1 (unit lower, transposed)   cpu.c:109-111 */
                for (int l = 0; l < j; l++) {
                    double s = p[l];
                    for (int r = l+1; r < j; r++) s += V[l*ld+r]*p[r];
                    w[l] = s;
                }
                /* w += V2^T p2                           cpu.c:114-115 */
                for (int l = 0; l < j; l++) {
                    double s = 0.0;
                    for (int r = j; r < m; r++) s += V[l*ld+r]*p[r];
                    w[l] += s;
                }
                /* w = T^T w (upper, transposed)          cpu.c:118-120 */
                for (int l = j-1; l >= 0; l--) {
                    double s = 0.0;
                    for (int r = 0; r <= l; r++) s += T[(size_t)l*nb+r]*w[r];
                    w[l] = s;
                }
                /* p2 -= V2 w                             cpu.c:123-124 */
                for (int l = 0; l < j; l++) {
                    double s = w[l];
                    for (int r = j; r < m; r++) p[r] -= V[l*ld+r]*s;
                }
                /* p1 -= V1 w (unit lower)                cpu.c:127-130 */
                for (int r = j-1; r >= 0; r--) {
                    double s = w[r];
                    for (int l = 0; l < r; l++) s += V[l*ld+r]*w[l];
                    p[r] -= s;
                }
                /* the workspace column is overwritten again by finish_column
                 * of the last column; keep T's strictly lower part clean */
                if (j < nb-1) for (int l = 0; l < j; l++) w[l] = 0.0;
            }
            /* reflector                                   cpu.c:137-160 */
            double *v = V+j*ld+j;
            memcpy(v, p+j, (m-j)*sizeof(double));
            double tau = oracle_dlarfg(m-j, p+j, v+1);
            v[0] = 1.0;
            for (int r = j+1; r < m; r++) p[r] = 0.0;
            T[(size_t)j*nb+j] = tau;

            /* ---- compute_column: y = A(i+1:end, i+j+1:end) v   cpu.c:217 ---- */
            #pragma omp parallel
            {
                #pragma omp for schedule(static)
                for (int rb = 0; rb < m; rb += 256) {
                    int re = MIN(m, rb+256);
                    for (int r = rb; r < re; r++) y[r] = 0.0;
                    for (int c = 0; c < m-j; c++) {
                        double s = v[c];
                        double const *a = A+(size_t)(i+j+1+c)*ldA+i+1;
                        for (int r = rb; r < re; r++) y[r] += a[r]*s;
                    }
                }
            }

            /* ---- finish_column, cpu.c:253-284 ---- */
            double *Yj = Y+j*ld, *Tj = T+(size_t)j*nb;
            memcpy(Yj, y, m*sizeof(double));
            for (int l = 0; l < j; l++) {                  /* cpu.c:263-264 */
                double s = 0.0;
                for (int r = j; r < m; r++) s += V[l*ld+r]*V[j*ld+r];
                Tj[l] = s;
            }
            for (int l = 0; l < j; l++) {                  /* cpu.c:267-268 */
                double s = Tj[l];
                double const *yl = Y+l*ld;
                for (int r = 0; r < m; r++) Yj[r] -= yl[r]*s;
            }
            for (int r = 0; r < m; r++) Yj[r] *= tau;      /* cpu.c:270 */
            for (int l = 0; l < j; l++) Tj[l] *= -tau;     /* cpu.c:277 */
            for (int l = 0; l < j; l++) {                  /* cpu.c:280-282 */
                double s = 0.0;
                for (int r = l; r < j; r++) s += T[(size_t)r*nb+l]*Tj[r];
                Tj[l] = s;
            }
            Tj[j] = tau;                                   /* cpu.c:284 */
        }

        /* ---- critical trailing updates, core.c:523-547 ---- */
        int const nt = end-(i+nb);     /* trailing columns i+nb .. end-1 */
        if (0 < nt) {
            double *At = A+(size_t)(i+nb)*ldA+i+1;
            /* A -= Y * V(nb-1:,:)^T                      cpu.c:315-316 */
            gemm_nt_sub(m, nt, nb, Y, ld, V+nb-1, ld, At, ldA);
            /* W = A^T V ; W = W T ; A -= V W^T           cpu.c:373-384,433 */
            double *W = malloc((size_t)nt*nb*sizeof(double));
            if (!W) return -1;
            gemm_tn(m, nt, nb, At, ldA, V, ld, W, nt);
            trmm_right_upper(nt, nb, T, nb, W, nt);
            gemm_nt_sub(m, nt, nb, V, ld, W, nt, At, ldA);
            free(W);
        }
        free(Y); free(y);

        struct panel *pn = malloc(sizeof *pn);               /* core.c:555 */
        if (!pn) return -1;
        pn->i = i; pn->nb = nb; pn->m = m; pn->P = P; pn->V = V; pn->T = T;
        pn->next = NULL;
        if (tail) tail->next = pn; else head = pn;
        tail = pn;
    }

    /* ---- delayed updates, core.c:301-349, in panel order ---- */
    for (struct panel *pn = head; pn != NULL; ) {
        int const i = pn->i, nb = pn->nb, m = pn->m;
        size_t const ld = m;
        for (int j = 0; j < nb; j++)                          /* core.c:317 */
            memcpy(A+(size_t)(i+j)*ldA+i+1, pn->P+j*ld, m*sizeof(double));
        {   /* A(0:i+1, i+1:end) (I - V T V^T)                core.c:321-327 */
            int const rows = i+1;
            double *W = malloc((size_t)rows*nb*sizeof(double));
            if (!W) return -1;
            double *Ar = A+(size_t)(i+1)*ldA;
            gemm_nn(rows, m, nb, Ar, ldA, pn->V, ld, W, rows);
            trmm_right_upper(rows, nb, pn->T, nb, W, rows);
            gemm_nt_sub(rows, m, nb, W, rows, pn->V, ld, Ar, ldA);
            free(W);
        }
        if (end < n) {   /* columns right of end               core.c:330-336 */
            int const nt = n-end;
            double *At = A+(size_t)end*ldA+i+1;
            double *W = malloc((size_t)nt*nb*sizeof(double));
            if (!W) return -1;
            gemm_tn(m, nt, nb, At, ldA, pn->V, ld, W, nt);
            trmm_right_upper(nt, nb, pn->T, nb, W, nt);
            gemm_nt_sub(m, nt, nb, pn->V, ld, W, nt, At, ldA);
            free(W);
        }
        {   /* Q(:, i+1:end) (I - V T V^T)                     core.c:339-340 */
            double *W = malloc((size_t)n*nb*sizeof(double));
            if (!W) return -1;
            double *Qr = Q+(size_t)(i+1)*ldQ;
            gemm_nn(n, m, nb, Qr, ldQ, pn->V, ld, W, n);
            trmm_right_upper(n, nb, pn->T, nb, W, n);
            gemm_nt_sub(n, m, nb, W, n, pn->V, ld, Qr, ldQ);
            free(W);
        }
        struct panel *nx = pn->next;
        free(pn->P); free(pn->V); free(pn->T); free(pn);
        pn = nx;
    }
    return 0;
}

/* hessenberg/interface.c:74-78 */
int oracle_default_panel_width(int n)
{
    int a = (int)(0.001875596476*n + 273.5908216);   /* divceil(int,int): common.h:207 */
    int w = (a+7)/8*8;
    return MAX(64, w);
}

/* ------------------------------------------------------------------------
 * The reference's checks, test/common/checks.c:180-208 and
 * test/common/hooks.c:434-456, in units of u = 2^-52.
 * ---------------------------------------------------------------------- */

/* 2^52 * ||Q H Q^T - A||_F / ||A||_F */
double oracle_residual_u(int n, double const *Q, int ldQ, double const *H,
    int ldH, double const *A, int ldA)
{
    double *X = malloc((size_t)n*n*sizeof(double));
    double *R = malloc((size_t)n*n*sizeof(double));
    if (!X || !R) return -1.0;
    /* X = Q H */
    #pragma omp parallel for schedule(static)
    for (int j = 0; j < n; j++) {
        double *x = X+(size_t)j*n;
        for (int i = 0; i < n; i++) x[i] = 0.0;
        for (int l = 0; l < n; l++) {
            double h = H[(size_t)j*ldH+l];
            if (h == 0.0) continue;
            double const *q = Q+(size_t)l*ldQ;
            for (int i = 0; i < n; i++) x[i] += q[i]*h;
        }
    }
    /* R = X Q^T - A */
    #pragma omp parallel for schedule(static)
    for (int j = 0; j < n; j++) {
        double *r = R+(size_t)j*n;
        for (int i = 0; i < n; i++) r[i] = -A[(size_t)j*ldA+i];
        for (int l = 0; l < n; l++) {
            double q = Q[(size_t)l*ldQ+j];
            double const *x = X+(size_t)l*n;
            for (int i = 0; i < n; i++) r[i] += x[i]*q;
        }
    }
    double num = 0.0, den = 0.0;
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) {
            double r = R[(size_t)j*n+i], a = A[(size_t)j*ldA+i];
            num += r*r; den += a*a;
        }
    free(X); free(R);
    return ldexp(sqrt(num)/sqrt(den), 52);
}

/* 2^52 * ||Q Q^T - I||_F / sqrt(n) */
double oracle_orthogonality_u(int n, double const *Q, int ldQ)
{
    double num = 0.0;
    #pragma omp parallel for schedule(static) reduction(+:num)
    for (int j = 0; j < n; j++) {
        double *r = calloc(n, sizeof(double));
        for (int l = 0; l < n; l++) {
            double q = Q[(size_t)l*ldQ+j];
            double const *x = Q+(size_t)l*ldQ;
            for (int i = 0; i < n; i++) r[i] += x[i]*q;
        }
        r[j] -= 1.0;
        for (int i = 0; i < n; i++) num += r[i]*r[i];
        free(r);
    }
    return ldexp(sqrt(num)/sqrt((double)n), 52);
}

/* number of non-zero entries with row >= col+2 (must be 0) */
long oracle_count_below_subdiagonal(int n, double const *H, int ldH)
{
    long cnt = 0;
    for (int j = 0; j < n; j++)
        for (int i = j+2; i < n; i++)
            if (H[(size_t)j*ldH+i] != 0.0) cnt++;
    return cnt;
}
