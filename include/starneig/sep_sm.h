/* Standard eigenvalue problem, shared memory (one node).  Replaces: reference
 * src/include/starneig/sep_sm.h:89-92 (Hessenberg), :380-384 (expert variant); the Schur entry points
 * (:126-130, :424-429) are added with the Schur path.  All arrays are HOST pointers,
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

#ifdef __cplusplus
}
#endif
#endif
