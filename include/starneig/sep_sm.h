/* Standard eigenvalue problem, shared memory (one node).  Replaces: reference
 * src/include/starneig/sep_sm.h:89-92 (Hessenberg), :126-130 (Schur),
 * :380-384 and :424-429 (expert variants).  All arrays are HOST pointers,
 * column-major, results are written in place, exactly as in the reference;
 * the library moves them to HBM, runs the HIP path and copies back. */
#ifndef STARNEIG_AMD_SEP_SM_H
#define STARNEIG_AMD_SEP_SM_H
#include <starneig/error.h>
#include <starneig/expert.h>
#ifdef __cplusplus
extern "C" {
#endif

/* A <- H (exact zeros below the sub-diagonal), Q <- Q*U, Q*H*Q^T = A_in.
 * Errors: n<1 -> -1, A NULL -> -2, ldA<n -> -3, Q NULL -> -4, ldQ<n -> -5
 * (hessenberg/interface.c:175-179); STARNEIG_NOT_INITIALIZED. */
starneig_error_t starneig_SEP_SM_Hessenberg(
    int n, double A[], int ldA, double Q[], int ldQ);

/* Reduce columns [begin,end) only (sep_sm.h:353-357).  Errors shifted by the
 * leading conf argument: n<1 -> -2, begin<0 -> -3, n<end -> -4, A NULL -> -5,
 * ldA<n -> -6, Q NULL -> -7, ldQ<n -> -8 (hessenberg/interface.c:144-150). */
starneig_error_t starneig_SEP_SM_Hessenberg_expert(
    struct starneig_hessenberg_conf *conf, int n, int begin, int end,
    double A[], int ldA, double Q[], int ldQ);

/* H (upper Hessenberg) <- S (real Schur form, 2x2 blocks standardised),
 * Q <- Q*U; real/imag receive the eigenvalues in diagonal order (may be NULL).
 * Errors: n<1 -> -1, H NULL -> -2, ldH<n -> -3, Q NULL -> -4, ldQ<n -> -5
 * (schur/interface.c:228-232); STARNEIG_DID_NOT_CONVERGE. */
starneig_error_t starneig_SEP_SM_Schur(
    int n, double H[], int ldH, double Q[], int ldQ,
    double real[], double imag[]);

/* Errors: n<1 -> -2, H NULL -> -3, ldH<n -> -4, Q NULL -> -5, ldQ<n -> -6
 * (schur/interface.c:198-202). */
starneig_error_t starneig_SEP_SM_Schur_expert(
    struct starneig_schur_conf *conf, int n, double H[], int ldH,
    double Q[], int ldQ, double real[], double imag[]);

/* Evaluates `predicate` on every eigenvalue of the Schur form S (host array); both members
 * of a complex pair get the same flag, a selected pair counts twice (reference
 * sep_sm.h:327-334, common/helpers.c:47-101).  Errors: n<1 -> -1, S NULL -> -2, ldS<n -> -3,
 * predicate NULL -> -4, selected NULL -> -6. */
starneig_error_t starneig_SEP_SM_Select(
    int n, double S[], int ldS,
    int (*predicate)(double real, double imag, void *arg), void *arg,
    int selected[], int *num_selected);

/* reference sep_sm.h:174-179, :474-480 (reorder/interface.c:210-263): moves the selected
 * eigenvalues (selected[i] != 0; a 2x2 block is selected as a whole) to the top-left corner of
 * the Schur form, S <- U^T S U, Q <- Q U; on exit selected[] marks the final positions of the
 * correctly placed eigenvalues.  Returns STARNEIG_PARTIAL_REORDERING when an exchange of two
 * blocks was rejected as numerically unstable (the decomposition stays valid). */
starneig_error_t starneig_SEP_SM_ReorderSchur(
    int n, int selected[], double S[], int ldS, double Q[], int ldQ,
    double real[], double imag[]);

starneig_error_t starneig_SEP_SM_ReorderSchur_expert(
    struct starneig_reorder_conf *conf, int n, int selected[],
    double S[], int ldS, double Q[], int ldQ, double real[], double imag[]);

/* reference sep_sm.h:232-240 (common/combined.c:46-98): Hessenberg + Schur; with a predicate
 * also Select + ReorderSchur, the sequence of examples/sep_sm_full_chain.c:88-121 (selected /
 * num_selected are then written).  predicate may be NULL (no reordering).  Errors: n<1 -> -1,
 * A NULL -> -2, ldA<n -> -3, Q NULL -> -4, ldQ<n -> -5. */
starneig_error_t starneig_SEP_SM_Reduce(
    int n, double A[], int ldA, double Q[], int ldQ,
    double real[], double imag[],
    int (*predicate)(double real, double imag, void *arg), void *arg,
    int selected[], int *num_selected);

#ifdef __cplusplus
}
#endif
#endif
