/* Return codes of the starneig_* entry points.
 * Replaces: reference src/include/starneig/error.h (codes 0..8) and the
 * "-k = k-th argument invalid" convention of hessenberg/interface.c:144-150,
 * :175-179 and schur/interface.c:198-202, :228-232. */
#ifndef STARNEIG_AMD_ERROR_H
#define STARNEIG_AMD_ERROR_H
#ifdef __cplusplus
extern "C" {
#endif

typedef int starneig_error_t;

#define STARNEIG_SUCCESS                 0
#define STARNEIG_GENERIC_ERROR           1
#define STARNEIG_NOT_INITIALIZED         2
#define STARNEIG_INVALID_CONFIGURATION   3
#define STARNEIG_INVALID_ARGUMENTS       4
#define STARNEIG_INVALID_DISTR_MATRIX    5
#define STARNEIG_DID_NOT_CONVERGE        6
#define STARNEIG_PARTIAL_REORDERING      7
#define STARNEIG_CLOSE_EIGENVALUES       8

#ifdef __cplusplus
}
#endif
#endif
