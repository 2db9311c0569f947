// Shared declarations of the MI355X (gfx950) Hessenberg/Schur path.
// Everything here is device-side plumbing behind the C-ABI of include/starneig/*.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define SN_HIP_CHECK(expr)                                                      \
    do {                                                                        \
        hipError_t e_ = (expr);                                                 \
        if (e_ != hipSuccess) {                                                 \
            fprintf(stderr, "[starneig-amd] HIP error %s at %s:%d: %s\n",       \
                hipGetErrorName(e_), __FILE__, __LINE__, hipGetErrorString(e_));\
            abort();                                                            \
        }                                                                       \
    } while (0)

namespace sn {

static inline int divceil(int a, int b) { return (a + b - 1) / b; }
static inline size_t roundup(size_t a, size_t b) { return (a + b - 1) / b * b; }

// ---- fp64 MFMA GEMM (dgemm_mfma.hip) ---------------------------------------
// C(m x n) = alpha * op(A) * op(B) + beta * C, column-major, BLAS semantics.
// transA/transB: 'N' or 'T'.  beta == 0 never reads C.
void dgemm(hipStream_t s, char transA, char transB, int m, int n, int k,
    double alpha, double const *A, int lda, double const *B, int ldb,
    double beta, double *C, int ldc);

void dgemm_accurate(hipStream_t s, char transA, char transB, int m, int n, int k,
    double const *A, int lda, double const *B, int ldb, double *C, int ldc);

struct GemmDesc {               // one problem of a batched launch (device pointers)
    double const *A; double const *B; double *C;
    int m, n, k, lda, ldb, ldc;
};
void dgemm_batched_left_inplace(hipStream_t s, GemmDesc const *ddescs, int count, int max_cols);
void dgemm_batched_right_inplace(hipStream_t s, GemmDesc const *ddescs, int count, int max_rows);
void dgemm_left_inplace(hipStream_t s, int w, int ncols, double const *U, int ldu,
    double *X, int ldx);
void dgemm_right_inplace(hipStream_t s, int nrows, int w, double const *U, int ldu,
    double *X, int ldx);

// ---- small helpers (util.hip) -----------------------------------------------
void make_stream(hipStream_t *s, bool critical, int prio, int free_cus = 0);
void copy_matrix(hipStream_t s, int m, int n, double const *A, int lda,
    double *B, int ldb);
void set_matrix(hipStream_t s, int m, int n, double value, double diag,
    double *A, int lda);
// Fill with the reference test driver's LCG (test/common/common.c:56-59),
// element e (column-major order, ld skipped) = state after e+1 steps.
// mode 0: prand/PRAND_MAX, mode 1: 2*prand/PRAND_MAX-1.
void lcg_fill(hipStream_t s, int m, int n, unsigned seed, int mode,
    double *A, int lda);

// ---- Hessenberg (hessenberg.hip) --------------------------------------------
struct HessenbergTimings {
    int sample_every = 0;       // in: time every k-th gemv launch with HIP events (0 = off)
    float total_ms = 0.f;       // whole reduction, event-timed on the caller's stream
    double gemv_bytes = 0.0;    // algorithmic bytes streamed by all panel gemv launches
    double gemm_flops = 0.0;    // executed GEMM flops (all updates)
    double gemm_flops_main = 0.0;   // of which: the critical trailing update (rows H4-H6)
    double gemm_ms_main = 0.0;      // its summed duration (events on the critical stream)
    double gemm_ms_side = 0.0;      // summed duration of the delayed updates (Q, upper rows) on the side stream
    double gemm_flops_fused = 0.0;  // of gemm_flops_main: the fused k = 2 nb update A -= [Y V][V' W]^T alone
    double gemm_ms_fused = 0.0;     // its summed duration
    long gemv_launches = 0;
    long sampled_launches = 0;
    double sampled_bytes = 0.0; // algorithmic bytes of the sampled launches
    double sampled_ms = 0.0;    // their summed kernel durations
    // sharded reduction only (SURVEY 8d "scaling report"): the collectives by kind, event-timed on the
    // reduction's stream -- [0] per-column all-reduce of y (the SAMPLED columns only), [1] per-panel
    // broadcast of the panel columns, [2] per-panel all-reduce of W (rows above the panel), [3] assembly of
    // H and Q at the end (broadcasts, one per block column / row block chunk)
    double comm_ms[4] = {0, 0, 0, 0};
    double comm_bytes[4] = {0, 0, 0, 0};    // payload bytes of the timed calls (count * 8)
    long comm_calls[4] = {0, 0, 0, 0};      // timed calls (for [0]: the sampled ones)
    long allreduce_y_calls = 0;             // all per-column all-reduces issued
};

// Reduces columns [begin,end) of the device-resident n x n matrix dA (ld ldA)
// and accumulates dQ <- dQ*U.  Blocking w.r.t. the given stream is left to the
// caller: all work is enqueued on ctx streams and joined back into `s`.
int hessenberg_device(hipStream_t s, int n, int begin, int end, int panel_width,
    double *dA, int ldA, double *dQ, int ldQ, HessenbergTimings *timings);

// Device-side exchange of the per-column partial vectors y = A_r v between the ranks of a one-process team
// (node_team.hip): rank r's gemv launch writes its folded row tiles into slot r of EVERY rank's buffer (peer
// stores: the same device, or peers over xGMI) and raises a flag per tile; the next column kernel of each rank
// waits for the flags of the tiles it reads and sums the world slots in rank order -- no host round trip, no
// collective launch, the same bits on every rank.  Passed to the kernels by value.
constexpr int HESS_MAX_RANKS = 16;
constexpr int HESS_MAX_ROW_TILES = 256;     // row tiles of a gemv launch (512 rows each): n <= 131072
struct HessExchange {
    double *slots[HESS_MAX_RANKS];  // slots[r]: rank r's buffer, [2 parities][world][ldp] doubles
    int *flags[HESS_MAX_RANKS];     // flags[r]: rank r's flags, [world][HESS_MAX_ROW_TILES]: last sequence number published
    int *error;                     // this rank's error word (a wait that timed out)
    int world, rank;
    int seq_base;                   // sequence number of the last column of the previous reduction
};

// Collective callbacks of the sharded reduction.  buffer ids: 0 = y vector, 1 = panel P,
// 2 = W scratch, 3 = A, 4 = Q (all allocated by the caller, who maps the id to its handle).
struct HessComm {
    int rank, world;
    void (*allreduce_sum)(void *ctx, int buffer, long offset, long count);
    void (*broadcast)(void *ctx, int buffer, long offset, long count, int root);
    void *ctx;
    HessExchange const *exchange = nullptr;     // non-null: the per-column all-reduce of y runs on the device (above)
};
int hessenberg_sharded_device(hipStream_t s, int n, int panel_width,
    double *dA, int ldA, double *dQ, int ldQ,
    double *dYsum, double *dP, double *dW2, long w2_capacity,
    HessComm const &comm, HessenbergTimings *timings);
int hessenberg_panel_ld(int n, int panel_width);
// RCCL called directly (rccl_native.hip; librccl.so opened at run time)
int rccl_unique_id(void *id128);
int rccl_init(int rank, int world, void const *id128);
void rccl_finalize();
bool rccl_ready(int rank, int world);
int rccl_comm_count();
int rccl_allreduce_sum(double *buf, long count, hipStream_t s);
int rccl_broadcast(double *buf, long count, int root, hipStream_t s);

// ---- Schur (schur.hip) -----------------------------------------------------------
struct SchurParams {            // resolved from starneig_schur_conf (negative = default)
    int iteration_limit = -1;
    int small_limit = -1;
    int aed_window_size = -1;
    int aed_nibble = -1;
    int shift_count = -1;
    double threshold = -1.0;    // -1/-2: norm-stable (u*||H||_F), -3: LAPACK criterion, >0: absolute
    int shifts_per_window = -1; // caps the shifts (2 x bulges) of one chain
    int aed_parallel_hard_limit = -1;   // AED windows up to this size use the sequential host kernel
    double threshold_b = -1.0;  // pencils: deflation threshold of B (right_threshold)
    double threshold_inf = -1.0;// pencils: infinite-eigenvalue threshold (inf_threshold)
    int host_threads = 1;       // starneig_node_init's `cores`: >= 6 gives the host window kernel its helper team (schur_host_team.h)
    // several GPUs reduce replicas of H (q_rows < n): rank r of `shard_world` alone keeps the 128-column
    // tiles T of the deflated part with T % shard_world == r up to date; the caller assembles H from them
    int shard_rank = 0, shard_world = 1;
};
struct SchurStats {
    int sweeps = 0, aeds = 0, small_solves = 0, chase_launches = 0;
    double gemm_flops = 0.0;
    float total_ms = 0.f;
    double aed_host_s = 0.0;    // wall time inside the host AED kernel
    double wait_s = 0.0;        // wall time the host waited for the GPU (window downloads)
    int inf_deflated = 0;       // pencils: infinite eigenvalues chased out and deflated
};
// Reduces the device-resident upper Hessenberg matrix dH to real Schur form, dQ <- dQ*U.
// real/imag are HOST arrays (may be NULL).  Returns a starneig_error_t value.
// q_rows >= 0: dQ points at a block of q_rows rows of Q and only those are updated (row-sharded
// accumulation of Q over several GPUs that each reduce a replica of H).
int schur_device(hipStream_t s, int n, double *dH, int ldH, double *dQ, int ldQ,
    double *real, double *imag, SchurParams const &params, SchurStats *stats, int q_rows = -1,
    int level = 0);
void schur_release_workspace();
// Generalized twin (schur_gep.hip): (dA, dB) Hessenberg-triangular -> generalized Schur form
int gep_schur_device(hipStream_t s, int n, double *dA, int ldA, double *dB, int ldB,
    double *dQ, int ldQ, double *dZ, int ldZ, double *real, double *imag, double *beta,
    SchurParams const &params, SchurStats *stats, int level = 0);
void gep_schur_release_workspace();
void lcg_pencil(hipStream_t s, int n, unsigned seed, double *H, int ldh, double *R, int ldr);
// Hessenberg-triangular reduction (hess_tri.hip): general (dA, dB) -> (H, T), dQ <- dQ*U1, dZ <- dZ*U2
int hessenberg_triangular_device(hipStream_t s, int n, double *dA, int ldA, double *dB, int ldB,
    double *dQ, int ldQ, double *dZ, int ldZ, double *stats);
void hessenberg_triangular_release_workspace();

} // namespace sn
