/* Device-pointer entry points of the MI355X path (extension of the StarNEig
 * interface).  The starneig_SEP_SM_* functions of include/starneig/sep_sm.h take
 * HOST arrays like the reference does (hessenberg/interface.c:138-167 registers
 * the caller's arrays in place); these take arrays that are already resident in
 * HBM -- what a caller that keeps its data on the GPU (or a benchmark that must
 * not time PCIe) binds instead.  Plain pointers and sizes only; `stream` is a
 * hipStream_t passed as void* (NULL = default stream).
 * All matrices are column-major doubles.  Workspaces and internal streams are cached per
 * process: the entry points are not reentrant (one reduction at a time per process). */
#ifndef STARNEIG_AMD_H
#define STARNEIG_AMD_H
#include <starneig/error.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Replaces starneig_hessenberg_insert_tasks (hessenberg/core.c:351) + codelets
 * (hessenberg/cpu.c:50-560, cuda.cu:62-309) on device-resident data.
 * panel_width <= 0 selects the reference default (hessenberg/interface.c:74-78).
 * dQ may be NULL (Q not accumulated).  stats (may be NULL) is double[16]:
 * in  [7] = k > 0: time every k-th panel-gemv launch with HIP events (0 = off);
 * out [0] total ms (events on `stream`), [1] algorithmic bytes of all panel-gemv
 * launches, [2] executed GEMM flops, [3] summed duration (ms) and [4] algorithmic
 * bytes of the sampled gemv launches, [5] gemv launches, [6] sampled launches,
 * [8] summed duration (ms, HIP events on the critical stream) and [9] executed flops of
 * the trailing-matrix updates (rows H4-H6), [10] summed duration (ms) of the delayed
 * updates of Q and the upper rows on the side stream ([2] - [9] flops).
 * Blocks until the result is complete. */
starneig_error_t starneig_amd_hessenberg_device(
    int n, int begin, int end, int panel_width,
    double *dA, int ldA, double *dQ, int ldQ, void *stream, double *stats);

/* Block-column sharded Hessenberg reduction over the GPUs of one node, one process per
 * GPU (the reference's distributed counterpart is the StarPU-MPI path,
 * hessenberg/tasks.c:337-345, mpi/interface_hessenberg.c:150-195).  Every rank passes
 * full-size dA/dQ holding the same input and gets the full result back.  The caller owns
 * the communication: it allocates dY (ld doubles), dP (ld*panel_width) and dW (w_capacity
 * >= n*panel_width doubles), ld = starneig_amd_hessenberg_panel_ld(n, panel_width), and
 * supplies allreduce_sum / broadcast callbacks that act on (buffer, offset, count) with
 * buffer 0 = dY, 1 = dP, 2 = dW, 3 = dA, 4 = dQ, ordered on `stream`
 * (torch.distributed over RCCL in starneig_amd/distributed.py).  With BOTH callbacks NULL the
 * collectives are issued to RCCL directly on `stream` (no host round trip per panel column), through
 * the communicator of starneig_amd_rccl_init.
 * stats (may be NULL) is ALWAYS double[32], zero-initialised by the caller (it is read on entry): out [0] total
 * ms, [1] algorithmic bytes this rank's gemv launches streamed, [2] executed GEMM flops, [5] gemv launches.
 * In: stats[7] = k, an integer in [1, 2^20] (anything else -- NaN, a fraction, out of range -- counts as 0 =
 * off): every k-th gemv launch (with the all-reduce behind it) and every per-panel collective is
 * timed with HIP events on `stream` (scaling report): [8] ms, [9] bytes, [10] count of the
 * sampled launches; [11] per-column all-reduces issued; [12 + 3i .. 14 + 3i] ms, payload bytes, timed calls
 * of collective kind i = 0 all-reduce of y (sampled columns), 1 panel broadcast, 2 all-reduce of W, 3 assembly
 * of H and Q; [24] ranks of the library's RCCL communicator as ncclCommCount reports them (0: callbacks). */
int starneig_amd_hessenberg_panel_ld(int n, int panel_width);

/* RCCL called directly (librccl.so is opened at run time).  One communicator per process: rank 0
 * obtains the 128-byte id, the caller carries it to the other ranks (any transport), every rank
 * calls _init(rank, world, id).  _allreduce_sum / _broadcast act in place on device doubles, ordered
 * on `stream`.  All return 0 on success. */
int starneig_amd_rccl_unique_id(void *id128);
int starneig_amd_rccl_init(int rank, int world, void const *id128);
void starneig_amd_rccl_finalize(void);
int starneig_amd_rccl_allreduce_sum(double *dbuf, long count, void *stream);
int starneig_amd_rccl_broadcast(double *dbuf, long count, int root, void *stream);
starneig_error_t starneig_amd_hessenberg_sharded_device(
    int n, int panel_width, double *dA, int ldA, double *dQ, int ldQ,
    double *dY, double *dP, double *dW, long w_capacity,
    int rank, int world,
    void (*allreduce_sum)(void *ctx, int buffer, long offset, long count),
    void (*broadcast)(void *ctx, int buffer, long offset, long count, int root),
    void *ctx, void *stream, double *stats);

/* Replaces starneig_schur_insert_tasks (schur/core.c:2342-2514) + the window and update
 * codelets (schur/cpu.c, cpu_utils.c, common/cpu.c:54-162) on a device-resident upper
 * Hessenberg matrix: dH <- real Schur form, dQ <- dQ*U (dQ may be NULL).  real/imag are
 * HOST arrays of length n (both NULL = not extracted).  conf may be NULL (defaults).
 * stats (may be NULL) is double[8]: [0] total ms, [1] QR sweeps, [2] AED calls,
 * [3] small host solves, [4] chase launches, [5] executed GEMM flops, [6] seconds inside
 * the host AED kernel, [7] seconds the host waited for the GPU.
 * Returns STARNEIG_DID_NOT_CONVERGE like schur/core.c:2324-2326. */
struct starneig_schur_conf;
starneig_error_t starneig_amd_schur_device(
    int n, double *dH, int ldH, double *dQ, int ldQ, double *real, double *imag,
    struct starneig_schur_conf *conf, void *stream, double *stats);

/* Row-sharded accumulation of Q: every GPU of a node reduces its own replica of
 * H (the reduction is deterministic, the replicas stay bit-identical) but updates only the q_rows
 * rows of Q that start at dQrows (= dQ + first_row, same leading dimension) -- 45 % of the update
 * flops of the Schur leg shard this way without any communication; the caller assembles Q
 * afterwards (starneig_amd/distributed.py: zero the rows a rank does not own, all-reduce). */
starneig_error_t starneig_amd_schur_rows_device(
    int n, double *dH, int ldH, double *dQrows, int ldQ, int q_rows, double *real, double *imag,
    struct starneig_schur_conf *conf, void *stream, double *stats);

/* The same with the left updates of the DEFLATED columns of H sharded as well: those columns (right of
 * the active block) are never read or mixed by a later window, they only receive further left updates,
 * column by column.  Rank r of `world` keeps the 128-column tiles T (columns [128 T, 128 T + 128)) with
 * T % world == r up to date and skips the others -- still no communication during the reduction.  On
 * return column tile T of the Schur form is valid on rank T % world only (the diagonal blocks, the
 * eigenvalues and everything left of them are valid everywhere); the caller assembles H from the
 * owners' tiles (starneig_amd/distributed.py: one all-gather of the packed tiles; node_team.hip: every rank copies
 * its tiles back).  What the reference does by moving tiles between workers (schur/core.c:129-460
 * insert_updates over StarPU-MPI) is done here without moving anything until the end. */
starneig_error_t starneig_amd_schur_sharded_device(
    int n, double *dH, int ldH, double *dQrows, int ldQ, int q_rows, int rank, int world,
    double *real, double *imag, struct starneig_schur_conf *conf, void *stream, double *stats);

/* Eigenvalue reordering on a device-resident Schur form (replaces reorder/core.c + cpu.c /
 * cuda.cu:126-761 + the GEMM updates of common/cpu.c:54-162): dS <- U^T dS U, dQ <- dQ U (dQ may
 * be NULL).  selected is a HOST array (in: marks of the selected eigenvalues, out: final positions
 * of the placed ones); real/imag HOST arrays (both NULL = not extracted).  conf may be NULL.
 * stats (may be NULL) is double[4]: [0] windows processed, [1] executed GEMM flops, [2] rounds. */
struct starneig_reorder_conf;
starneig_error_t starneig_amd_reorder_schur_device(
    int n, int *selected, double *dS, int ldS, double *dQ, int ldQ, double *real, double *imag,
    struct starneig_reorder_conf *conf, void *stream, double *stats);

/* Generalized twin (BASELINE config 5): the device-resident Hessenberg-triangular pencil
 * (dH, dR) <- generalized real Schur form, dQ <- dQ*U1, dZ <- dZ*U2 (either may be NULL).
 * real/imag/beta are HOST arrays of length n (all NULL = not extracted).  Same conf and
 * stats layout as starneig_amd_schur_device.  Replaces the generalized branches of
 * schur/core.c:2342-2514 and schur/cpu_utils.c:1168-1810. */
starneig_error_t starneig_amd_gep_schur_device(
    int n, double *dH, int ldH, double *dR, int ldR, double *dQ, int ldQ, double *dZ, int ldZ,
    double *real, double *imag, double *beta,
    struct starneig_schur_conf *conf, void *stream, double *stats);

/* Device-resident twin of starneig_GEP_SM_HessenbergTriangular (wrappers/lapack.c:45-176):
 * general (dA, dB) -> (H, T); dQ / dZ may be NULL.  Below n = 1500 the rotations of LAPACK dgghrd (results agree
 * with dgeqrf + dormqr + dgghrd up to rounding), from there on a two-stage Householder reduction (band form by
 * blocked QR / RQ, then a bulge chase; backward stable, no LAPACK counterpart entry by entry).  stats (may be NULL,
 * double[8]): [0] total ms, [1] QR step ms, [2] ms of the step after it (rotations, or both stages), [3] executed
 * GEMM flops of the QR step, [4] rotations applied (0 on the two-stage path), [5] 1 = two-stage path, [6] its
 * stage 1 ms. */
starneig_error_t starneig_amd_hessenberg_triangular_device(
    int n, double *dA, int ldA, double *dB, int ldB, double *dQ, int ldQ, double *dZ, int ldZ,
    void *stream, double *stats);

/* fp64 MFMA GEMM, BLAS dgemm semantics on device pointers (the kernel behind
 * rows H4-H8 and S3; replaces cblas_dgemm/cublasDgemm call sites). */
starneig_error_t starneig_amd_dgemm_device(
    char transA, char transB, int m, int n, int k, double alpha,
    double const *dA, int ldA, double const *dB, int ldB, double beta,
    double *dC, int ldC, void *stream);

/* The reference test driver's LCG inputs generated directly in HBM
 * (test/common/common.c:56-59, test/common/init.c:93-120):
 * mode 0 = prand/PRAND_MAX in [0,1] ("fullpos"), mode 1 = 2*prand/PRAND_MAX-1. */
starneig_error_t starneig_amd_lcg_fill_device(
    int m, int n, unsigned seed, int mode, double *dA, int ldA, void *stream);

/* A <- value everywhere, diag on the diagonal. */
starneig_error_t starneig_amd_set_matrix_device(
    int m, int n, double value, double diag, double *dA, int ldA, void *stream);

/* The reference's acceptance checks evaluated on the GPU
 * (test/common/checks.c:180-208): out[0] = 2^52 ||Q H Q^T - A||_F / ||A||_F,
 * out[1] = 2^52 ||Q Q^T - I||_F / sqrt(n), out[2] = number of non-zero entries
 * below the first sub-diagonal of H (test/common/hooks.c:434-456).
 * dWork1/dWork2: n x n scratch (ld n). */
starneig_error_t starneig_amd_check_device(
    int n, double const *dQ, int ldQ, double const *dH, int ldH,
    double const *dA0, int ldA0, double *dWork1, double *dWork2,
    double out[3], void *stream);

/* The reference test driver's random Hessenberg-triangular pencil (test/schur/
 * experiment.c:203-207: generate_random_hessenberg then generate_random_uptriag on the one
 * LCG stream, test/common/init.c:122-138,159-175), generated directly in HBM. */
starneig_error_t starneig_amd_lcg_pencil_device(
    int n, unsigned seed, double *dH, int ldH, double *dR, int ldR, void *stream);

/* Acceptance checks of a two-sided decomposition on the GPU (test/common/checks.c):
 * out[0] = 2^52 ||Q S Z^T - A0||_F / ||A0||_F, out[1] = 2^52 ||Q Q^T - I||_F / sqrt(n),
 * out[2] = the same for Z, out[3] = non-zeros of S below the first sub-diagonal.
 * dWork1/dWork2 are n*n scratch. */
starneig_error_t starneig_amd_check_pencil_device(
    int n, double const *dQ, int ldQ, double const *dS, int ldS, double const *dZ, int ldZ,
    double const *dA0, int ldA0, double *dWork1, double *dWork2, double out[4], void *stream);

/* hessenberg/interface.c:74-78 */
int starneig_amd_default_panel_width(int n);

/* Frees cached device workspaces (they are otherwise kept between calls). */
void starneig_amd_release_workspace(void);

#ifdef __cplusplus
}
#endif
#endif
