// TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into or called by the product path).
//
// Restatement of the pieces of the reference's TEST DRIVER that build the inputs of its Schur
// experiments and judge their eigenvalues -- the only reference-held source of expected answers
// for the Schur path:
//   * `--init known`: a (generalized) Schur form with a prescribed spectrum, hidden behind random
//     Householder similarity transformations        test/schur/experiment.c:295-410
//   * the spectrum itself (`--complex-distr uniform`)  test/common/complex_distr.c:51-76,144-228
//   * placement of the 1x1 / 2x2 blocks               test/common/block_placer.c:53-100
//   * random Householder matrix                       test/common/init.c:173-191,523-541
//   * `--decouple k` / `--set-to-inf k`                test/schur/experiment.c:66-140
//   * the reorder experiment's input (`--fortify`) and `select_distr uniform`
//                                                     test/common/init_schur.c:122-181, select_distr.c:48-143
//   * the `known-eigenvalues` hook (greedy nearest match, warn 1e4 u, fail 1e6 u)
//                                                     test/common/hooks.c:1071-1296
//   * the `eigenvalues` hook (position by position, warn 1e3 u, fail 1e4 u)
//                                                     test/common/hooks.c:787-991
// All random draws go through the test driver's LCG (oracle_prand, test/common/common.c:51-59)
// in the reference's order, so that a given (n, seed) names the same experiment there and here.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PRAND_MAX 0x7fffffff
int oracle_prand(void);

static double squ(double x) { return x * x; }

// uniform_complex_distr_init (complex_distr.c:144-228) without --fortify, followed by
// generate_special_cases (:51-76).  `generalized` = "B != NULL" of the reference.
// fortify != 0: `--fortify` (complex_distr.c:165-171, :186-196) -- eigenvalues 2 apart, "fortified against
// failed swaps"; no special cases, no draws besides the placement of the 2x2 blocks.
void oracle_known_spectrum(int n, int generalized, double complex_ratio, double zero_ratio,
                           double inf_ratio, double *real, double *imag, double *beta, int fortify)
{
    int complex_count = complex_ratio * n / 2;
    int real_count = n - 2 * complex_count;
    int *spaces = calloc(complex_count + 1, sizeof(int));
    for (int i = 0; i < real_count; i++)
        spaces[oracle_prand() % (complex_count + 1)]++;
    if (fortify) {
        for (int i = 0; i < n; i++) { real[i] = 2.0 * (i - n / 2 + 0.5); imag[i] = 0.0; beta[i] = 1.0; }
        int i = 0;
        for (int j = 0; j < complex_count; j++) {
            i += spaces[j];
            imag[i] = fabs(real[i]);
            real[i + 1] = real[i];
            imag[i + 1] = -imag[i];
            i += 2;
        }
        free(spaces);
        return;
    }
    for (int i = 0; i < n; i++) {
        real[i] = (2.0 * oracle_prand() / PRAND_MAX - 1.0) * n;
        imag[i] = 0.0;
        beta[i] = generalized ? 1.0 * oracle_prand() / PRAND_MAX : 1.0;
    }
    int i = 0;
    for (int j = 0; j < complex_count; j++) {
        i += spaces[j];
        real[i] = (2.0 * oracle_prand() / PRAND_MAX - 1.0) * n;
        imag[i] = (2.0 * oracle_prand() / PRAND_MAX - 1.0) * n;
        real[i + 1] = real[i];
        imag[i + 1] = -imag[i];
        beta[i] = beta[i + 1] = 1.0 * oracle_prand() / PRAND_MAX;   // drawn in the standard case too
        i += 2;
    }
    free(spaces);
    // zero eigenvalues
    for (i = 0; i < n; i++) {
        if (i + 1 < n && imag[i] != 0.0)
            i++;
        else if (1.0 * oracle_prand() / PRAND_MAX < zero_ratio / (1.0 - complex_ratio))
            real[i] = 0.0;
    }
    // infinite eigenvalues (the reference's local beta array is never NULL: the draws happen in
    // the standard case as well; block_placer ignores beta there)
    for (i = 0; i < n; i++) {
        if (i + 1 < n && imag[i] != 0.0)
            i++;
        else if (real[i] != 0.0 &&
                 1.0 * oracle_prand() / PRAND_MAX < inf_ratio / (1.0 - complex_ratio - zero_ratio))
            beta[i] = 0.0;
    }
}

// block_placer.c:53-100 on one window covering the whole matrix.
void oracle_place_blocks(int n, const double *real, const double *imag, const double *beta,
                         double *A, int ldA, double *B, int ldB)
{
    int i = 0;
    while (i < n) {
        if (imag[i] != 0.0) {
            A[(size_t)i * ldA + i] = real[i];
            A[(size_t)(i + 1) * ldA + i + 1] = real[i + 1];
            A[(size_t)(i + 1) * ldA + i] = imag[i];
            A[(size_t)i * ldA + i + 1] = imag[i + 1];
            if (B) {
                B[(size_t)i * ldB + i] = beta[i];
                B[(size_t)(i + 1) * ldB + i + 1] = beta[i];
                B[(size_t)(i + 1) * ldB + i] = 0.0;
                B[(size_t)i * ldB + i + 1] = 0.0;
            }
            i += 2;
        } else {
            A[(size_t)i * ldA + i] = real[i];
            if (B)
                B[(size_t)i * ldB + i] = beta[i];
            i++;
        }
    }
}

// generate_random_householder (init.c:523-541): the unit vector v of Q = I - 2 v v^T.
void oracle_householder_vector(int n, double *v)
{
    for (int i = 0; i < n; i++)
        v[i] = 2.0 * (1.0 * oracle_prand() / PRAND_MAX) - 1.0;
    double scal = 0.0;
    for (int i = 0; i < n; i++)
        scal += v[i] * v[i];
    scal = 1.0 / sqrt(scal);
    for (int i = 0; i < n; i++)
        v[i] *= scal;
}

// deflate_and_place_infinities (schur/experiment.c:100-140 with the crawler at :66-88).
void oracle_decouple(int n, int cuts, int infinities, double *A, int ldA, double *B, int ldB)
{
    int *sub = calloc(n, sizeof(int));
    if (cuts > n - 1) cuts = n - 1;
    for (int i = 0; i < cuts; i++) {
        int p = oracle_prand() % (n - 1) + 1;
        while (sub[p] & 1)
            p = oracle_prand() % (n - 1) + 1;
        sub[p] |= 1;
    }
    for (int i = 0; i < infinities; i++) {
        int p = oracle_prand() % (n - 1) + 1;
        while (sub[p] & 2)
            p = oracle_prand() % (n - 1) + 1;
        sub[p] |= 2;
    }
    for (int i = 1; i < n; i++) {
        if (sub[i] & 1)
            A[(size_t)(i - 1) * ldA + i] = 0.0;
        if (B && (sub[i] & 2))
            B[(size_t)i * ldB + i] = 0.0;
    }
    free(sub);
}

// known_eigenvalues_test_after_solver_run (hooks.c:1178-1296).  (real1, imag1, beta1): computed;
// (real2, imag2, beta2): prescribed.  out = {mean, min, max}; counts = {warnings, failures}.
void oracle_known_eigenvalues_check(int n, const double *real1, const double *imag1,
                                    const double *beta1, const double *real2, const double *imag2,
                                    const double *beta2, double warn_threshold,
                                    double fail_threshold, double *out, int *counts)
{
    const double two52 = (double)((long long)1 << 52);
    int *used = calloc(n, sizeof(int));
    double mean = 0.0, mn = INFINITY, mx = 0.0;
    counts[0] = counts[1] = 0;
    for (int i = 0; i < n; i++) {
        int closest = n;
        double closest_diff = INFINITY;
        for (int j = 0; j < n; j++) {
            if (used[j]) continue;
            if (beta1[i] == 0.0 && beta2[j] == 0.0) {
                closest = j;
                closest_diff = 0.0;
                break;
            }
            double diff;
            if (real2[j] == 0.0 && imag2[j] == 0.0)
                diff = two52 * sqrt(squ(real1[i] / beta1[i]) + squ(imag1[i] / beta1[i]));
            else
                diff = two52 * sqrt(squ(real1[i] / beta1[i] - real2[j] / beta2[j]) +
                                    squ(imag1[i] / beta1[i] - imag2[j] / beta2[j])) /
                       sqrt(squ(real2[j] / beta2[j]) + squ(imag2[j] / beta2[j]));
            if (diff < closest_diff) {
                closest = j;
                closest_diff = diff;
            }
        }
        if (closest < n)
            used[closest] = 1;
        if (fail_threshold < closest_diff || isnan(closest_diff))
            counts[1]++;
        else if (warn_threshold < closest_diff)
            counts[0]++;
        mean += closest_diff;
        if (closest_diff < mn) mn = closest_diff;
        if (closest_diff > mx) mx = closest_diff;
    }
    free(used);
    out[0] = mean / n;
    out[1] = mn;
    out[2] = mx;
}

// eigenvalues_test_after_solver_run (hooks.c:891-991).  (real1, imag1, beta1): extracted from the
// diagonal blocks of the result; (real2, imag2, beta2): returned by the solver.
void oracle_eigenvalues_check(int n, const double *real1, const double *imag1, const double *beta1,
                              const double *real2, const double *imag2, const double *beta2,
                              double warn_threshold, double fail_threshold, double *out,
                              int *counts)
{
    const double two52 = (double)((long long)1 << 52);
    double mean = 0.0, mn = INFINITY, mx = 0.0;
    counts[0] = counts[1] = 0;
    for (int i = 0; i < n; i++) {
        if (real1[i] == 0.0 && imag1[i] == 0.0 && real2[i] == 0.0 && imag2[i] == 0.0)
            continue;
        if (beta1[i] == 0.0 && beta2[i] == 0.0)
            continue;
        double diff;
        if (beta1[i] == 0.0 && beta2[i] != 0.0) {
            diff = INFINITY;
            counts[1]++;
        } else {
            diff = two52 *
                   sqrt(squ(real1[i] / beta1[i] - real2[i] / beta2[i]) +
                        squ(imag1[i] / beta1[i] - imag2[i] / beta2[i])) /
                   sqrt(squ(real1[i] / beta1[i]) + squ(imag1[i] / beta1[i]));
            if (fail_threshold < diff || isnan(diff))
                counts[1]++;
            else if (warn_threshold < diff)
                counts[0]++;
        }
        mean += diff;
        if (diff < mn) mn = diff;
        if (diff > mx) mx = diff;
    }
    out[0] = mean / n;
    out[1] = mn;
    out[2] = mx;
}

// uniform_select_distr_init (test/common/select_distr.c:48-60, :122-143): whole diagonal blocks drawn with
// the LCG until `ratio` of the rows is selected.  A: the (quasi-)triangular matrix.
void oracle_uniform_select(int n, const double *A, int ldA, double ratio, int *select)
{
    int *blocks = calloc(n, sizeof(int));
    for (int i = 1; i < n; i++) blocks[i] = A[(size_t)(i - 1) * ldA + i] != 0.0;
    for (int i = 0; i < n; i++) select[i] = 0;
    int selected = 0;
    while (selected < ratio * n) {
        int const i = oracle_prand() % n;
        if (select[i] || blocks[i]) continue;
        select[i] = 1; selected++;
        if (i + 1 < n && blocks[i + 1]) { select[i + 1] = 1; selected++; }
    }
    free(blocks);
}
