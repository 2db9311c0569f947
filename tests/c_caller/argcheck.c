/* A plain C caller compiled against include/starneig/starneig.h and linked with
 * -lstarneig_amd: the argument-check table of the reference interface
 * (hessenberg/interface.c:144-150,175-182; schur/interface.c:198-202,228-232,250-258,
 * 286-294; common/combined.c:57-61; common/helpers.c:55-62).  Runs WITHOUT a GPU: every
 * check precedes the STARNEIG_NOT_INITIALIZED test, and the node is never initialised. */
#include <stdio.h>
#include <stdlib.h>
#include <starneig/starneig.h>

static int failures = 0;
#define EXPECT(expr, want) do { \
    int got_ = (expr); \
    if (got_ != (want)) { printf("FAIL %s = %d, expected %d\n", #expr, got_, (want)); failures++; } \
} while (0)

static int always(double re, double im, void *arg) { (void)re; (void)im; (void)arg; return 1; }

int main(void)
{
    enum { n = 5, ld = 8 };
    double *A = calloc(ld * n, sizeof(double)), *Q = calloc(ld * n, sizeof(double));
    double *B = calloc(ld * n, sizeof(double)), *Z = calloc(ld * n, sizeof(double));
    double re[n], im[n], be[n];
    int sel[n], cnt = 0;

    EXPECT(starneig_node_initialized(), 0);

    EXPECT(starneig_SEP_SM_Hessenberg(0, A, ld, Q, ld), -1);
    EXPECT(starneig_SEP_SM_Hessenberg(n, NULL, ld, Q, ld), -2);
    EXPECT(starneig_SEP_SM_Hessenberg(n, A, n - 1, Q, ld), -3);
    EXPECT(starneig_SEP_SM_Hessenberg(n, A, ld, NULL, ld), -4);
    EXPECT(starneig_SEP_SM_Hessenberg(n, A, ld, Q, n - 1), -5);
    EXPECT(starneig_SEP_SM_Hessenberg(n, A, ld, Q, ld), STARNEIG_NOT_INITIALIZED);

    struct starneig_hessenberg_conf hc;
    starneig_hessenberg_init_conf(&hc);
    EXPECT(hc.tile_size, STARNEIG_HESSENBERG_DEFAULT_TILE_SIZE);
    EXPECT(hc.panel_width, STARNEIG_HESSENBERG_DEFAULT_PANEL_WIDTH);
    EXPECT(starneig_SEP_SM_Hessenberg_expert(&hc, 0, 0, n, A, ld, Q, ld), -2);
    EXPECT(starneig_SEP_SM_Hessenberg_expert(&hc, n, -1, n, A, ld, Q, ld), -3);
    EXPECT(starneig_SEP_SM_Hessenberg_expert(&hc, n, 0, n + 1, A, ld, Q, ld), -4);
    EXPECT(starneig_SEP_SM_Hessenberg_expert(&hc, n, 0, n, NULL, ld, Q, ld), -5);
    EXPECT(starneig_SEP_SM_Hessenberg_expert(&hc, n, 0, n, A, n - 1, Q, ld), -6);
    EXPECT(starneig_SEP_SM_Hessenberg_expert(&hc, n, 0, n, A, ld, NULL, ld), -7);
    EXPECT(starneig_SEP_SM_Hessenberg_expert(&hc, n, 0, n, A, ld, Q, n - 1), -8);
    EXPECT(starneig_SEP_SM_Hessenberg_expert(NULL, n, 0, n, A, ld, Q, ld), STARNEIG_NOT_INITIALIZED);

    EXPECT(starneig_SEP_SM_Schur(0, A, ld, Q, ld, re, im), -1);
    EXPECT(starneig_SEP_SM_Schur(n, NULL, ld, Q, ld, re, im), -2);
    EXPECT(starneig_SEP_SM_Schur(n, A, n - 1, Q, ld, re, im), -3);
    EXPECT(starneig_SEP_SM_Schur(n, A, ld, NULL, ld, re, im), -4);
    EXPECT(starneig_SEP_SM_Schur(n, A, ld, Q, n - 1, re, im), -5);
    /* real / imag are not argument-checked (NULL = eigenvalues not extracted, schur/core.c:2501) */
    EXPECT(starneig_SEP_SM_Schur(n, A, ld, Q, ld, NULL, NULL), STARNEIG_NOT_INITIALIZED);

    struct starneig_schur_conf sc;
    starneig_schur_init_conf(&sc);
    EXPECT(sc.iteration_limit, STARNEIG_SCHUR_DEFAULT_INTERATION_LIMIT);
    EXPECT(sc.window_size, STARNEIG_SCHUR_DEFAULT_WINDOW_SIZE);
    EXPECT(sc.left_threshold == STARNEIG_SCHUR_DEFAULT_THRESHOLD, 1);
    EXPECT(sc.inf_threshold == STARNEIG_SCHUR_DEFAULT_THRESHOLD, 1);
    EXPECT(starneig_SEP_SM_Schur_expert(&sc, 0, A, ld, Q, ld, re, im), -2);
    EXPECT(starneig_SEP_SM_Schur_expert(&sc, n, NULL, ld, Q, ld, re, im), -3);
    EXPECT(starneig_SEP_SM_Schur_expert(&sc, n, A, n - 1, Q, ld, re, im), -4);
    EXPECT(starneig_SEP_SM_Schur_expert(&sc, n, A, ld, NULL, ld, re, im), -5);
    EXPECT(starneig_SEP_SM_Schur_expert(&sc, n, A, ld, Q, n - 1, re, im), -6);
    EXPECT(starneig_SEP_SM_Schur_expert(&sc, n, A, ld, Q, ld, re, im), STARNEIG_NOT_INITIALIZED);

    EXPECT(starneig_GEP_SM_Schur(0, A, ld, B, ld, Q, ld, Z, ld, re, im, be), -1);
    EXPECT(starneig_GEP_SM_Schur(n, NULL, ld, B, ld, Q, ld, Z, ld, re, im, be), -2);
    EXPECT(starneig_GEP_SM_Schur(n, A, n - 1, B, ld, Q, ld, Z, ld, re, im, be), -3);
    EXPECT(starneig_GEP_SM_Schur(n, A, ld, NULL, ld, Q, ld, Z, ld, re, im, be), -4);
    EXPECT(starneig_GEP_SM_Schur(n, A, ld, B, n - 1, Q, ld, Z, ld, re, im, be), -5);
    EXPECT(starneig_GEP_SM_Schur(n, A, ld, B, ld, NULL, ld, Z, ld, re, im, be), -6);
    EXPECT(starneig_GEP_SM_Schur(n, A, ld, B, ld, Q, n - 1, Z, ld, re, im, be), -7);
    EXPECT(starneig_GEP_SM_Schur(n, A, ld, B, ld, Q, ld, NULL, ld, re, im, be), -8);
    EXPECT(starneig_GEP_SM_Schur(n, A, ld, B, ld, Q, ld, Z, n - 1, re, im, be), -9);
    /* nothing beyond -9: NULL eigenvalue arrays reach the init check */
    EXPECT(starneig_GEP_SM_Schur(n, A, ld, B, ld, Q, ld, Z, ld, NULL, NULL, NULL), STARNEIG_NOT_INITIALIZED);
    EXPECT(starneig_GEP_SM_Schur_expert(&sc, 0, A, ld, B, ld, Q, ld, Z, ld, re, im, be), -2);
    EXPECT(starneig_GEP_SM_Schur_expert(&sc, n, A, ld, B, ld, Q, ld, Z, n - 1, re, im, be), -10);
    EXPECT(starneig_GEP_SM_Schur_expert(&sc, n, A, ld, B, ld, Q, ld, Z, ld, NULL, NULL, NULL), STARNEIG_NOT_INITIALIZED);

    /* wrappers/lapack.c:65-73, common/combined.c:110-118 */
    EXPECT(starneig_GEP_SM_HessenbergTriangular(0, A, ld, B, ld, Q, ld, Z, ld), -1);
    EXPECT(starneig_GEP_SM_HessenbergTriangular(n, NULL, ld, B, ld, Q, ld, Z, ld), -2);
    EXPECT(starneig_GEP_SM_HessenbergTriangular(n, A, n - 1, B, ld, Q, ld, Z, ld), -3);
    EXPECT(starneig_GEP_SM_HessenbergTriangular(n, A, ld, NULL, ld, Q, ld, Z, ld), -4);
    EXPECT(starneig_GEP_SM_HessenbergTriangular(n, A, ld, B, n - 1, Q, ld, Z, ld), -5);
    EXPECT(starneig_GEP_SM_HessenbergTriangular(n, A, ld, B, ld, NULL, ld, Z, ld), -6);
    EXPECT(starneig_GEP_SM_HessenbergTriangular(n, A, ld, B, ld, Q, n - 1, Z, ld), -7);
    EXPECT(starneig_GEP_SM_HessenbergTriangular(n, A, ld, B, ld, Q, ld, NULL, ld), -8);
    EXPECT(starneig_GEP_SM_HessenbergTriangular(n, A, ld, B, ld, Q, ld, Z, n - 1), -9);
    EXPECT(starneig_GEP_SM_HessenbergTriangular(n, A, ld, B, ld, Q, ld, Z, ld), STARNEIG_NOT_INITIALIZED);
    EXPECT(starneig_GEP_SM_Reduce(0, A, ld, B, ld, Q, ld, Z, ld, re, im, be, NULL, NULL, NULL, NULL), -1);
    EXPECT(starneig_GEP_SM_Reduce(n, A, ld, B, ld, Q, ld, Z, n - 1, re, im, be, NULL, NULL, NULL, NULL), -9);
    EXPECT(starneig_GEP_SM_Reduce(n, A, ld, B, ld, Q, ld, Z, ld, re, im, be, NULL, NULL, NULL, NULL), STARNEIG_NOT_INITIALIZED);

    EXPECT(starneig_SEP_SM_Reduce(0, A, ld, Q, ld, re, im, NULL, NULL, NULL, NULL), -1);
    EXPECT(starneig_SEP_SM_Reduce(n, NULL, ld, Q, ld, re, im, NULL, NULL, NULL, NULL), -2);
    EXPECT(starneig_SEP_SM_Reduce(n, A, n - 1, Q, ld, re, im, NULL, NULL, NULL, NULL), -3);
    EXPECT(starneig_SEP_SM_Reduce(n, A, ld, NULL, ld, re, im, NULL, NULL, NULL, NULL), -4);
    EXPECT(starneig_SEP_SM_Reduce(n, A, ld, Q, n - 1, re, im, NULL, NULL, NULL, NULL), -5);
    /* nothing beyond -5 (common/combined.c:57-61) */
    EXPECT(starneig_SEP_SM_Reduce(n, A, ld, Q, ld, NULL, NULL, NULL, NULL, NULL, NULL), STARNEIG_NOT_INITIALIZED);

    EXPECT(starneig_SEP_SM_Select(0, A, ld, always, NULL, sel, &cnt), -1);
    EXPECT(starneig_SEP_SM_Select(n, NULL, ld, always, NULL, sel, &cnt), -2);
    EXPECT(starneig_SEP_SM_Select(n, A, n - 1, always, NULL, sel, &cnt), -3);
    EXPECT(starneig_SEP_SM_Select(n, A, ld, NULL, NULL, sel, &cnt), -4);
    EXPECT(starneig_SEP_SM_Select(n, A, ld, always, NULL, NULL, &cnt), -6);
    EXPECT(starneig_SEP_SM_Select(n, A, ld, always, NULL, sel, &cnt), STARNEIG_NOT_INITIALIZED);

    /* reorder/interface.c:213-218, :244-249 */
    EXPECT(starneig_SEP_SM_ReorderSchur(0, sel, A, ld, Q, ld, re, im), -1);
    EXPECT(starneig_SEP_SM_ReorderSchur(n, NULL, A, ld, Q, ld, re, im), -2);
    EXPECT(starneig_SEP_SM_ReorderSchur(n, sel, NULL, ld, Q, ld, re, im), -3);
    EXPECT(starneig_SEP_SM_ReorderSchur(n, sel, A, n - 1, Q, ld, re, im), -4);
    EXPECT(starneig_SEP_SM_ReorderSchur(n, sel, A, ld, NULL, ld, re, im), -5);
    EXPECT(starneig_SEP_SM_ReorderSchur(n, sel, A, ld, Q, n - 1, re, im), -6);
    EXPECT(starneig_SEP_SM_ReorderSchur(n, sel, A, ld, Q, ld, re, im), STARNEIG_NOT_INITIALIZED);
    struct starneig_reorder_conf rc_;
    starneig_reorder_init_conf(&rc_);
    EXPECT(rc_.plan, STARNEIG_REORDER_DEFAULT_PLAN);
    EXPECT(rc_.window_size, STARNEIG_REORDER_DEFAULT_WINDOW_SIZE);
    EXPECT(starneig_SEP_SM_ReorderSchur_expert(&rc_, 0, sel, A, ld, Q, ld, re, im), -2);
    EXPECT(starneig_SEP_SM_ReorderSchur_expert(&rc_, n, sel, A, ld, Q, n - 1, re, im), -7);

    free(A); free(Q); free(B); free(Z);
    if (failures == 0) printf("argcheck ok\n");
    return failures ? 1 : 0;
}
