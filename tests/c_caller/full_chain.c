/* A plain C program that COMPUTES on the GPU through the C interface alone: the chain of the reference's
 * examples/sep_sm_full_chain.c:55-134 -- node init, Hessenberg reduction, Schur reduction, selection by a
 * predicate, reordering, node finalize -- followed by that example's acceptance checks
 * (examples/validate.c:63-130: 2^52 ||Q S Q^T - C||_F / ||C||_F and 2^52 ||Q Q^T - I||_F / sqrt(n), both
 * below 1000).  Compiled with `gcc -I include ... -lstarneig_amd`, no Python, no torch, no BLAS: the matrix
 * products of the checks are plain loops.  The input comes from a 64-bit LCG instead of rand() so that a run
 * can be repeated; n is the first argument (default 1000).
 *
 * Beyond the example: the Schur form is checked entry by entry (zeros below the sub-diagonal, no two
 * consecutive sub-diagonal entries), the returned eigenvalues against the diagonal blocks, and the
 * selected eigenvalues must occupy the leading positions after the reordering. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <starneig/starneig.h>

static int predicate(double real, double imag, void *arg)
{
    (void)imag; (void)arg;
    return 0.0 < real;
}

static uint64_t lcg_state = 2019;
static double lcg_uniform(void)
{
    lcg_state = lcg_state * 6364136223846793005ULL + 1442695040888963407ULL;
    return 2.0 * (double)(lcg_state >> 11) / 9007199254740992.0 - 1.0;
}

/* C = op(A) * op(B) for column-major n x n matrices; jki order so that the inner loop is contiguous */
static void mm(int n, int ta, int tb, double const *A, size_t ldA, double const *B, size_t ldB, double *C, size_t ldC)
{
    for (int j = 0; j < n; j++) {
        double *c = C + j * ldC;
        for (int i = 0; i < n; i++) c[i] = 0.0;
        for (int k = 0; k < n; k++) {
            double const b = tb ? B[k * ldB + j] : B[j * ldB + k];
            if (b == 0.0) continue;
            if (!ta) {
                double const *a = A + k * ldA;
                for (int i = 0; i < n; i++) c[i] += a[i] * b;
            } else {
                for (int i = 0; i < n; i++) c[i] += A[i * ldA + k] * b;
            }
        }
    }
}

static int fail(char const *what, double value)
{
    fprintf(stderr, "full_chain FAILED: %s (%g)\n", what, value);
    return EXIT_FAILURE;
}

int main(int argc, char **argv)
{
    int const n = argc > 1 ? atoi(argv[1]) : 1000;
    if (n < 2) return fail("n", n);
    size_t const ld = ((size_t)n / 8 + 1) * 8;       /* the example's leading dimension */

    double *A = malloc(n * ld * sizeof(double)), *C = malloc(n * ld * sizeof(double));
    double *Q = malloc(n * ld * sizeof(double));
    double *real = malloc(n * sizeof(double)), *imag = malloc(n * sizeof(double));
    int *select = malloc(n * sizeof(int));
    if (!A || !C || !Q || !real || !imag || !select) return fail("malloc", 0);
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) {
            A[j * ld + i] = C[j * ld + i] = lcg_uniform();
            Q[j * ld + i] = i == j ? 1.0 : 0.0;
        }

    starneig_node_init(STARNEIG_USE_ALL, 1, STARNEIG_HINT_SM | STARNEIG_NO_MESSAGES);
    if (!starneig_node_initialized()) return fail("node not initialised", 0);
    if (starneig_node_get_gpus() != 1) return fail("gpus", starneig_node_get_gpus());

    int rc = starneig_SEP_SM_Hessenberg(n, A, (int)ld, Q, (int)ld);
    if (rc != STARNEIG_SUCCESS) return fail("starneig_SEP_SM_Hessenberg", rc);
    for (int j = 0; j < n; j++)
        for (int i = j + 2; i < n; i++)
            if (A[j * ld + i] != 0.0) return fail("entry below the sub-diagonal after the Hessenberg reduction", A[j * ld + i]);

    rc = starneig_SEP_SM_Schur(n, A, (int)ld, Q, (int)ld, real, imag);
    if (rc != STARNEIG_SUCCESS) return fail("starneig_SEP_SM_Schur", rc);

    int num_selected = -1;
    rc = starneig_SEP_SM_Select(n, A, (int)ld, &predicate, NULL, select, &num_selected);
    if (rc != STARNEIG_SUCCESS) return fail("starneig_SEP_SM_Select", rc);
    printf("Selected %d eigenvalues out of %d.\n", num_selected, n);
    if (num_selected <= 0 || num_selected >= n) return fail("num_selected", num_selected);

    rc = starneig_SEP_SM_ReorderSchur(n, select, A, (int)ld, Q, (int)ld, real, imag);
    if (rc != STARNEIG_SUCCESS) return fail("starneig_SEP_SM_ReorderSchur", rc);

    starneig_node_finalize();
    if (starneig_node_initialized()) return fail("node still initialised", 1);

    /* quasi-triangular form, eigenvalues of the diagonal blocks, selected ones first */
    for (int j = 0; j < n; j++)
        for (int i = j + 2; i < n; i++)
            if (A[j * ld + i] != 0.0) return fail("entry below the sub-diagonal of the Schur form", A[j * ld + i]);
    for (int i = 0; i + 2 < n; i++)
        if (A[i * ld + i + 1] != 0.0 && A[(i + 1) * ld + i + 2] != 0.0) return fail("two consecutive sub-diagonal entries", i);
    double anorm = 0.0;
    for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) anorm += C[j * ld + i] * C[j * ld + i];
    anorm = sqrt(anorm);
    for (int i = 0; i < n; ) {
        if (i + 1 < n && A[i * ld + i + 1] != 0.0) {
            double const a = A[i * ld + i], b = A[(i + 1) * ld + i], c = A[i * ld + i + 1], d = A[(i + 1) * ld + i + 1];
            if (a != d || b * c >= 0.0) return fail("2 x 2 block not in standard form", i);
            double const w = sqrt(fabs(b)) * sqrt(fabs(c));
            if (fabs(real[i] - a) + fabs(real[i + 1] - a) + fabs(imag[i] - w) + fabs(imag[i + 1] + w) > 1e-12 * anorm)
                return fail("eigenvalues of a 2 x 2 block", i);
            i += 2;
        } else {
            if (real[i] != A[i * ld + i] || imag[i] != 0.0) return fail("eigenvalue of a 1 x 1 block", i);
            i++;
        }
    }
    for (int i = 0; i < n; i++)
        if ((0.0 < real[i]) != (i < num_selected)) return fail("selected eigenvalues are not the leading ones", i);

    /* examples/validate.c: residual and orthogonality below 1000 u */
    double *T = malloc(n * ld * sizeof(double)), *Y = malloc(n * ld * sizeof(double));
    if (!T || !Y) return fail("malloc", 0);
    mm(n, 0, 0, Q, ld, A, ld, T, ld);
    mm(n, 0, 1, T, ld, Q, ld, Y, ld);
    double dot = 0.0;
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) { double const d = Y[j * ld + i] - C[j * ld + i]; dot += d * d; }
    double const residual = 4503599627370496.0 * sqrt(dot) / anorm;
    mm(n, 0, 1, Q, ld, Q, ld, T, ld);
    dot = 0.0;
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) { double const d = T[j * ld + i] - (i == j ? 1.0 : 0.0); dot += d * d; }
    double const orth = 4503599627370496.0 * sqrt(dot) / sqrt((double)n);
    printf("residual %.1f u, orthogonality %.1f u\n", residual, orth);
    if (!(residual < 1000.0)) return fail("The residual is too large", residual);
    if (!(orth < 1000.0)) return fail("Matrix is not orthogonal", orth);

    free(A); free(C); free(Q); free(real); free(imag); free(select); free(T); free(Y);
    printf("full_chain ok\n");
    return 0;
}
