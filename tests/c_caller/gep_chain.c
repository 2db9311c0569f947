/* A plain C program that computes the GENERALIZED chain on the GPU through the C interface alone: the part of the
 * reference's examples/gep_sm_full_chain.c:55-150 that this library provides -- node init, reduction of a dense pair
 * (A, B) to Hessenberg-triangular form, generalized Schur (QZ) reduction, node finalize (the generalized Select /
 * ReorderSchur that follow in the example are not built) -- followed by that example's acceptance
 * checks (examples/validate.c:63-130 for both matrices: 2^52 ||Q S Z^T - C||_F / ||C||_F, 2^52 ||Q Q^T - I||_F / sqrt(n),
 * the same for Z; all below 1000).  No Python, no torch, no BLAS: the products of the checks are plain loops.  LCG input
 * instead of rand(); n is the first argument (default 800).  Beyond the example: the generalized Schur form entry by
 * entry (S quasi-triangular, T triangular, 2 x 2 blocks of S over a diagonal block of T), and alpha / beta of every
 * 1 x 1 block against the diagonal entries. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <starneig/starneig.h>

static uint64_t lcg_state = 2019;
static double lcg_uniform(void)
{
    lcg_state = lcg_state * 6364136223846793005ULL + 1442695040888963407ULL;
    return 2.0 * (double)(lcg_state >> 11) / 9007199254740992.0 - 1.0;
}

/* C = op(A) * op(B), column-major n x n; the inner loop contiguous */
static void mm(int n, int ta, int tb, double const *A, size_t ldA, double const *B, size_t ldB, double *C, size_t ldC)
{
    for (int j = 0; j < n; j++) {
        double *c = C + j * ldC;
        for (int i = 0; i < n; i++) c[i] = 0.0;
        for (int k = 0; k < n; k++) {
            double const b = tb ? B[k * ldB + j] : B[j * ldB + k];
            if (b == 0.0) continue;
            if (!ta) { double const *a = A + k * ldA; for (int i = 0; i < n; i++) c[i] += a[i] * b; }
            else for (int i = 0; i < n; i++) c[i] += A[i * ldA + k] * b;
        }
    }
}

static int fail(char const *what, double value)
{
    fprintf(stderr, "gep_chain FAILED: %s (%g)\n", what, value);
    return EXIT_FAILURE;
}

static double residual_u(int n, size_t ld, double const *Q, double const *S, double const *Z, double const *C, double *T, double *Y)
{
    mm(n, 0, 0, Q, ld, S, ld, T, ld);
    mm(n, 0, 1, T, ld, Z, ld, Y, ld);
    double dot = 0.0, nrm = 0.0;
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) { double const d = Y[j * ld + i] - C[j * ld + i]; dot += d * d; nrm += C[j * ld + i] * C[j * ld + i]; }
    return 4503599627370496.0 * sqrt(dot) / sqrt(nrm);
}
static double orthogonality_u(int n, size_t ld, double const *Q, double *T)
{
    mm(n, 0, 1, Q, ld, Q, ld, T, ld);
    double dot = 0.0;
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) { double const d = T[j * ld + i] - (i == j ? 1.0 : 0.0); dot += d * d; }
    return 4503599627370496.0 * sqrt(dot) / sqrt((double)n);
}

int main(int argc, char **argv)
{
    int const n = argc > 1 ? atoi(argv[1]) : 800;
    if (n < 2) return fail("n", n);
    size_t const ld = ((size_t)n / 8 + 1) * 8;
    size_t const bytes = n * ld * sizeof(double);
    double *A = malloc(bytes), *B = malloc(bytes), *C = malloc(bytes), *D = malloc(bytes), *Q = malloc(bytes), *Z = malloc(bytes);
    double *real = malloc(n * sizeof(double)), *imag = malloc(n * sizeof(double)), *beta = malloc(n * sizeof(double));
    if (!A || !B || !C || !D || !Q || !Z || !real || !imag || !beta) return fail("malloc", 0);
    for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) A[j * ld + i] = C[j * ld + i] = lcg_uniform();
    for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) B[j * ld + i] = D[j * ld + i] = lcg_uniform();
    for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) Q[j * ld + i] = Z[j * ld + i] = i == j ? 1.0 : 0.0;

    starneig_node_init(STARNEIG_USE_ALL, 1, STARNEIG_HINT_SM | STARNEIG_AWAKE_WORKERS | STARNEIG_NO_MESSAGES);
    if (!starneig_node_initialized()) return fail("node not initialised", 0);

    int rc = starneig_GEP_SM_HessenbergTriangular(n, A, (int)ld, B, (int)ld, Q, (int)ld, Z, (int)ld);
    if (rc != STARNEIG_SUCCESS) return fail("starneig_GEP_SM_HessenbergTriangular", rc);
    for (int j = 0; j < n; j++) {
        for (int i = j + 2; i < n; i++) if (A[j * ld + i] != 0.0) return fail("H: entry below the sub-diagonal", A[j * ld + i]);
        for (int i = j + 1; i < n; i++) if (B[j * ld + i] != 0.0) return fail("T: entry below the diagonal", B[j * ld + i]);
    }
    rc = starneig_GEP_SM_Schur(n, A, (int)ld, B, (int)ld, Q, (int)ld, Z, (int)ld, real, imag, beta);
    if (rc != STARNEIG_SUCCESS) return fail("starneig_GEP_SM_Schur", rc);
    starneig_node_finalize();
    if (starneig_node_initialized()) return fail("node still initialised", 1);

    /* generalized Schur form */
    for (int j = 0; j < n; j++) {
        for (int i = j + 2; i < n; i++) if (A[j * ld + i] != 0.0) return fail("S: entry below the sub-diagonal", A[j * ld + i]);
        for (int i = j + 1; i < n; i++) if (B[j * ld + i] != 0.0) return fail("T: entry below the diagonal", B[j * ld + i]);
    }
    int blocks2 = 0, infinite = 0;
    for (int i = 0; i < n; ) {
        if (i + 1 < n && A[i * ld + i + 1] != 0.0) {
            if (i + 2 < n && A[(i + 1) * ld + i + 2] != 0.0) return fail("two consecutive sub-diagonal entries", i);
            if (B[(i + 1) * ld + i] != 0.0) return fail("T not diagonal under a 2 x 2 block of S", i);
            if (!(imag[i] > 0.0 && imag[i + 1] == -imag[i] && real[i] == real[i + 1] && beta[i] == beta[i + 1]))
                return fail("a 2 x 2 block without a conjugate pair of eigenvalues", i);
            blocks2++; i += 2;
        } else {
            if (imag[i] != 0.0) return fail("a 1 x 1 block with a complex eigenvalue", i);
            /* (alpha, beta) of a 1 x 1 block are the diagonal entries up to a common positive scale */
            double const s = A[i * ld + i], t = B[i * ld + i];
            if (fabs(real[i] * t - beta[i] * s) > 1e-10 * (fabs(real[i] * t) + fabs(beta[i] * s) + 1e-300)) return fail("alpha / beta of a 1 x 1 block", i);
            if (beta[i] == 0.0) infinite++;
            i++;
        }
    }
    printf("%d 2 x 2 blocks, %d infinite eigenvalues\n", blocks2, infinite);

    double *T = malloc(bytes), *Y = malloc(bytes);
    if (!T || !Y) return fail("malloc", 0);
    double const ra = residual_u(n, ld, Q, A, Z, C, T, Y), rb = residual_u(n, ld, Q, B, Z, D, T, Y);
    double const oq = orthogonality_u(n, ld, Q, T), oz = orthogonality_u(n, ld, Z, T);
    printf("residuals %.1f / %.1f u, orthogonality %.1f / %.1f u\n", ra, rb, oq, oz);
    if (!(ra < 1000.0) || !(rb < 1000.0)) return fail("The residual is too large", ra > rb ? ra : rb);
    if (!(oq < 1000.0) || !(oz < 1000.0)) return fail("Matrix is not orthogonal", oq > oz ? oq : oz);
    free(A); free(B); free(C); free(D); free(Q); free(Z); free(real); free(imag); free(beta); free(T); free(Y);
    printf("gep_chain ok\n");
    return 0;
}
