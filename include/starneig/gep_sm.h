/* Generalized eigenvalue problem, shared memory (one node).  Replaces: reference
 * src/include/starneig/gep_sm.h:106-111 (HessenbergTriangular), :164-170 (Schur), :316-326
 * (Reduce) and :503-510 (expert variant).  The Schur (QZ) leg is BASELINE config 5 (the input is
 * already a Hessenberg-triangular pencil); the Hessenberg-triangular reduction is the step before
 * it; _ReorderSchur and _Eigenvectors are outside.  All arrays are HOST pointers,
 * column-major, results are written in place, exactly as in the reference. */
#ifndef STARNEIG_AMD_GEP_SM_H
#define STARNEIG_AMD_GEP_SM_H
#include <starneig/error.h>
#include <starneig/expert.h>
#ifdef __cplusplus
extern "C" {
#endif

/* General (A, B) <- (H, T) upper Hessenberg / upper triangular, Q <- Q*U1, Z <- Z*U2 so that
 * Q (H,T) Z^T = Q_in (A,B) Z_in^T.  Same steps as the reference's LAPACK sequence
 * (wrappers/lapack.c:143-163: QR of B, then the rotation-based reduction), on the GPU.
 * Errors: n<1 -> -1, A NULL -> -2, ldA<n -> -3, B NULL -> -4, ldB<n -> -5, Q NULL -> -6,
 * ldQ<n -> -7, Z NULL -> -8, ldZ<n -> -9 (lapack.c:65-73); STARNEIG_NOT_INITIALIZED. */
starneig_error_t starneig_GEP_SM_HessenbergTriangular(
    int n, double A[], int ldA, double B[], int ldB,
    double Q[], int ldQ, double Z[], int ldZ);

/* HessenbergTriangular followed by Schur (common/combined.c:98-153).  The reference also reorders
 * when a predicate is given; the generalized reordering is not part of this library and a
 * non-NULL predicate returns STARNEIG_GENERIC_ERROR. */
starneig_error_t starneig_GEP_SM_Reduce(
    int n, double A[], int ldA, double B[], int ldB,
    double Q[], int ldQ, double Z[], int ldZ,
    double real[], double imag[], double beta[],
    int (*predicate)(double real, double imag, double beta, void *arg), void *arg,
    int selected[], int *num_selected);

/* (H, R) upper Hessenberg / upper triangular <- (S, T) generalized real Schur form
 * (S quasi-triangular, T upper triangular, 2x2 blocks standardised: T diagonal and
 * non-negative there), Q <- Q*U1, Z <- Z*U2 so that Q (S,T) Z^T = Q_in (H,R) Z_in^T.
 * The eigenvalues are (real[i] + i*imag[i]) / beta[i] in diagonal order.
 * Errors: n<1 -> -1, H NULL -> -2, ldH<n -> -3, R NULL -> -4, ldR<n -> -5, Q NULL -> -6,
 * ldQ<n -> -7, Z NULL -> -8, ldZ<n -> -9 (schur/interface.c:283-291; the checks stop there:
 * NULL real / imag / beta mean "eigenvalues not extracted", as in the reference);
 * STARNEIG_NOT_INITIALIZED; STARNEIG_DID_NOT_CONVERGE.  conf->right_threshold is honoured: it is
 * the magnitude below which an entry of R is negligible in the window kernels (LAPACK dhgeqz's BTOL);
 * a diagonal entry below it at the bottom of an active block is split off as an infinite
 * eigenvalue with beta = 0. */
starneig_error_t starneig_GEP_SM_Schur(
    int n, double H[], int ldH, double R[], int ldR,
    double Q[], int ldQ, double Z[], int ldZ,
    double real[], double imag[], double beta[]);

/* Same with a configuration structure; argument numbers in the error codes shift by one
 * (schur/interface.c:247-255). */
starneig_error_t starneig_GEP_SM_Schur_expert(
    struct starneig_schur_conf *conf, int n, double H[], int ldH, double R[], int ldR,
    double Q[], int ldQ, double Z[], int ldZ,
    double real[], double imag[], double beta[]);

#ifdef __cplusplus
}
#endif
#endif
