// C-ABI of the MI355X path: the StarNEig shared-memory interface for the
// Hessenberg/Schur hot path (include/starneig/*.h) plus the device-pointer
// extension (include/starneig_amd.h).  There is deliberately no CPU fallback:
// without a usable gfx950 device starneig_node_init() aborts.
#include "common.h"
#include "tuning.h"
#include <chrono>
#include "schur_host.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include <sched.h>
#include <starneig/starneig.h>
#include <starneig_amd.h>

namespace sn {
void hessenberg_release_workspace();
void schur_release_workspace();
void sumsq_diff(hipStream_t s, int m, int n, double const *X, int ldx, double const *Y, int ldy,
    double ident, double *acc);
void count_below(hipStream_t s, int n, double const *H, int ldh, double *acc);
int reorder_schur_device(hipStream_t caller, int n, int *selected, double *dS, int ldS,
    double *dQ, int ldQ, double *real, double *imag, int window_size, int values_per_chain,
    double *stats, int host_threads);
void reorder_release_workspace();
void upload_host_matrix(double *dev, int ldd, double const *host, int ldh, int rows, int cols, int threads);
void download_host_matrix(double *host, int ldh, double const *dev, int ldd, int rows, int cols, int threads);
void staging_release();
// several GPUs from one process (node_team.hip)
void node_team_start(int const *devices, int world);
void node_team_stop();
int node_team_world();
int node_team_hessenberg(int n, int panel_width, double *A, int ldA, double *Q, int ldQ, int cores);
int node_team_schur(int n, double *H, int ldH, double *Q, int ldQ, double *real, double *imag,
    SchurParams const &params, int cores);
}

namespace {

struct NodeState {                      // reference: static state, common/node.c:61-92
    bool initialized = false;
    int cores = 0, gpus = 0, avail_cores = 1;
    bool messages = true, verbose = true, pinning = false;
    int device = 0, avail_gpus = 1;
} g_node;

// gpus of starneig_node_init / starneig_node_set_gpus -> ranks of the in-process multi-GPU path
// (common/node.c:200-216: the reference hands min(requested, present) CUDA devices to StarPU).
// STARNEIG_AMD_VIRTUAL_GPUS=k (testing on a box with fewer devices): k ranks dealt round-robin over
// the devices present, collectives through the in-process exchange.
void configure_gpus(int gpus)
{
    int ndev = 1;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) ndev = 1;
    int avail = ndev;
    if (char const *v = getenv("STARNEIG_AMD_VIRTUAL_GPUS")) avail = std::max(1, std::min(16, atoi(v)));
    g_node.avail_gpus = avail;
    int const want = gpus == STARNEIG_USE_ALL ? avail : std::max(1, std::min(gpus, avail));
    sn::node_team_stop();
    g_node.gpus = want;
    if (want > 1) {
        std::vector<int> devs(want);
        for (int r = 0; r < want; r++) devs[r] = (g_node.device + r) % ndev;
        sn::node_team_start(devs.data(), want);
        if (sn::node_team_world() != want) g_node.gpus = 1;
    }
}

void require_device()
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count < 1) {
        fprintf(stderr, "[starneig-amd] fatal: no HIP device available (%s). "
            "This library has no CPU path.\n", hipGetErrorString(e));
        abort();
    }
    hipDeviceProp_t prop;
    SN_HIP_CHECK(hipGetDevice(&g_node.device));
    SN_HIP_CHECK(hipGetDeviceProperties(&prop, g_node.device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 && g_node.messages)
        fprintf(stderr, "[starneig-amd] warning: device is %s, kernels are built for gfx950\n",
            prop.gcnArchName);
}

// caller's (pageable) array <-> HBM, complete on return (staging.hip)
void to_device(double *dev, int ldd, double const *host, int ldh, int n)
{
    SN_HIP_CHECK(hipStreamSynchronize(nullptr));        // the buffer was cleared on the NULL stream
    sn::upload_host_matrix(dev, ldd, host, ldh, n, n, g_node.cores);
}
void to_host(double *host, int ldh, double const *dev, int ldd, int n)
{
    sn::download_host_matrix(host, ldh, dev, ldd, n, n, g_node.cores);
}

// A problem that cannot fit the device is an error code, not an abort deep inside an allocation: `count`
// n x n matrices of the caller plus the cached workspaces of the reduction (a generous 35 % of two
// matrices plus 1 GB) against what the device has free right now, the library's own caches included.
bool fits_device(int n, int count, char const *what)
{
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return true;
    double const mat = (double)sn::roundup(n, 16) * n * sizeof(double);
    double const need = count * mat + 0.7 * mat + 1e9;
    if (need <= (double)free_b) return true;
    if (g_node.messages)
        fprintf(stderr, "[starneig-amd] %s: n = %d needs about %.1f GB of device memory, %.1f GB are free\n",
            what, n, need / 1e9, (double)free_b / 1e9);
    return false;
}

// The panel width the library picks when the caller asks for the default (conf->panel_width =
// STARNEIG_HESSENBERG_DEFAULT_PANEL_WIDTH; the plain interface).  The reference's own choice is
// MAX(64, divceil((int)(0.001875596476 n + 273.59), 8) * 8) = 280 / 288 / 312 for n = 2000 / 8000 / 20000
// (hessenberg/interface.c:74-78), a fit to ITS tile kernels.  Here the per-column chain costs a pass over the
// panel factors that grows with the column's index in the panel, while the compact-WY updates lose little at
// k = 128 ... 192, so narrower panels are faster at every size measured on an MI355X (round 6,
// profiles/r6_panel_width_sweep.txt: n = 2000 0.057 -> 0.051 s, 4000 0.139 -> 0.127, 8000 0.486 -> 0.455, 12000
// 1.223 -> 1.173, 20000 4.66 -> 4.55 s; multiples of 64 sit best with the 128-wide GEMM tiles).  A caller who
// wants the reference's width passes it in the conf; every width from 8 up is honoured as before.
int default_panel_width(int n)
{
    return n <= 16000 ? 128 : 192;
}

} // namespace

extern "C" {

#define SN_API __attribute__((visibility("default")))

SN_API void starneig_node_init(int cores, int gpus, starneig_flag_t flags)
{
    if (g_node.initialized) {           // common/node.c:442-443: fatal
        fprintf(stderr, "[starneig-amd] fatal: the node is already initialized.\n");
        abort();
    }
    g_node.verbose = !(flags & STARNEIG_NO_VERBOSE);
    g_node.messages = (flags & STARNEIG_NO_MESSAGES) != STARNEIG_NO_MESSAGES;
    require_device();
    // cores: host threads available to the sequential window kernels (the process' affinity
    // mask, like the hwloc binding mask of common/node.c:497-536); gpus: the devices of this node that
    // the calls of this process use (configure_gpus above: one host thread per device); 0 is
    // refused -- there is no CPU path to fall back to.
    cpu_set_t mask;
    int avail = 1;
    if (sched_getaffinity(0, sizeof mask, &mask) == 0) avail = std::max(1, CPU_COUNT(&mask));
    // a CPU-time quota of the control group (a container started with --cpus=k keeps the full mask): the
    // helper threads of the window kernels spin, so threads beyond the quota would only get throttled
    {
        long quota = -1, period = -1;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                       // cgroup v2: "max 100000" or "200000 100000"
            char q[32] = {0};
            if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atol(q);
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
            if (fscanf(g, "%ld", &quota) != 1) quota = -1;
            fclose(g);
            if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%ld", &period) != 1) period = -1; fclose(h); }
        }
        if (quota > 0 && period > 0) avail = std::max(1, std::min(avail, (int)(quota / period)));
    }
    g_node.avail_cores = avail;
    g_node.cores = cores == STARNEIG_USE_ALL ? avail : std::max(1, std::min(cores, avail));
    if (gpus == 0) {
        fprintf(stderr, "[starneig-amd] fatal: starneig_node_init(gpus = 0): this library has no "
            "CPU path.\n");
        abort();
    }
    configure_gpus(gpus);
    g_node.initialized = true;
}

SN_API int starneig_node_initialized(void) { return g_node.initialized ? 1 : 0; }
SN_API int starneig_node_get_cores(void) { return g_node.cores; }
SN_API void starneig_node_set_cores(int cores)
{
    g_node.cores = cores == STARNEIG_USE_ALL ? g_node.avail_cores
                                             : std::max(1, std::min(cores, g_node.avail_cores));
}
SN_API int starneig_node_get_gpus(void) { return g_node.gpus; }
SN_API void starneig_node_set_gpus(int gpus)
{
    if (gpus == 0) {
        if (g_node.messages) fprintf(stderr, "[starneig-amd] warning: starneig_node_set_gpus(0) ignored: no CPU path.\n");
        return;
    }
    if (g_node.initialized) configure_gpus(gpus);
}
SN_API void starneig_node_enable_pinning(void) { g_node.pinning = true; }
SN_API void starneig_node_disable_pinning(void) { g_node.pinning = false; }

SN_API void starneig_node_finalize(void)
{
    if (!g_node.initialized) return;
    sn::node_team_stop();
    SN_HIP_CHECK(hipDeviceSynchronize());
    sn::hessenberg_release_workspace();
    sn::schur_release_workspace();
    sn::gep_schur_release_workspace();
    sn::reorder_release_workspace();
    sn::hessenberg_triangular_release_workspace();
    sn::staging_release();
    g_node.initialized = false;
}

SN_API void starneig_hessenberg_init_conf(struct starneig_hessenberg_conf *conf)
{
    conf->tile_size = STARNEIG_HESSENBERG_DEFAULT_TILE_SIZE;
    conf->panel_width = STARNEIG_HESSENBERG_DEFAULT_PANEL_WIDTH;
}

SN_API void starneig_schur_init_conf(struct starneig_schur_conf *conf)
{
    // every field -1 = "choose the default" (schur/interface.c:167-188)
    static const struct starneig_schur_conf all_default = {
        -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1.0, -1.0, -1.0 };
    *conf = all_default;
}

SN_API int starneig_amd_default_panel_width(int n) { return default_panel_width(n); }

SN_API void starneig_amd_release_workspace(void)
{
    sn::hessenberg_release_workspace();
    sn::schur_release_workspace();
    sn::gep_schur_release_workspace();
    sn::hessenberg_triangular_release_workspace();
}

// ---- host-array interface (in place, like the reference) ----------------------

SN_API starneig_error_t starneig_SEP_SM_Hessenberg_expert(
    struct starneig_hessenberg_conf *conf, int n, int begin, int end,
    double A[], int ldA, double Q[], int ldQ)
{
    if (n < 1)      return -2;           // hessenberg/interface.c:144-150
    if (begin < 0)  return -3;
    if (n < end)    return -4;
    if (A == NULL)  return -5;
    if (ldA < n)    return -6;
    if (Q == NULL)  return -7;
    if (ldQ < n)    return -8;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;

    int panel_width = STARNEIG_HESSENBERG_DEFAULT_PANEL_WIDTH;
    if (conf != NULL) {
        if (conf->tile_size != STARNEIG_HESSENBERG_DEFAULT_TILE_SIZE && conf->tile_size < 8)
            return STARNEIG_INVALID_CONFIGURATION;          // interface.c:68-71
        panel_width = conf->panel_width;
        if (panel_width != STARNEIG_HESSENBERG_DEFAULT_PANEL_WIDTH && panel_width < 8)
            return STARNEIG_INVALID_CONFIGURATION;          // interface.c:80-83
    }
    if (panel_width == STARNEIG_HESSENBERG_DEFAULT_PANEL_WIDTH)
        panel_width = default_panel_width(n);

    // several GPUs (starneig_node_init(cores, gpus > 1, ...)): the block-column sharded reduction, one
    // host thread per device (node_team.hip).  Partial ranges and matrices too small to shard stay on
    // one device.
    if (!fits_device(n, 2, "starneig_SEP_SM_Hessenberg")) return STARNEIG_GENERIC_ERROR;
    if (g_node.gpus > 1 && begin == 0 && end == n && n >= 256 * g_node.gpus)
        return sn::node_team_hessenberg(n, panel_width, A, ldA, Q, ldQ, g_node.cores) == 0
            ? STARNEIG_SUCCESS : STARNEIG_GENERIC_ERROR;

    int const ld = (int)sn::roundup(n, 16);
    size_t const bytes = (size_t)ld * n * sizeof(double);
    double *dA = nullptr, *dQ = nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tt[8]; tt[0] = now();
    SN_HIP_CHECK(hipMalloc((void **)&dA, bytes));
    SN_HIP_CHECK(hipMalloc((void **)&dQ, bytes));
    SN_HIP_CHECK(hipMemset(dA, 0, bytes));
    SN_HIP_CHECK(hipMemset(dQ, 0, bytes));
    SN_HIP_CHECK(hipStreamSynchronize(nullptr)); tt[1] = now();
    to_device(dA, ld, A, ldA, n); tt[2] = now();
    to_device(dQ, ld, Q, ldQ, n); tt[3] = now();

    int rc = sn::hessenberg_device(nullptr, n, begin, end, panel_width, dA, ld, dQ, ld, nullptr);
    SN_HIP_CHECK(hipStreamSynchronize(nullptr)); tt[4] = now();

    to_host(A, ldA, dA, ld, n); tt[5] = now();
    to_host(Q, ldQ, dQ, ld, n); tt[6] = now();
    SN_HIP_CHECK(hipFree(dA));
    SN_HIP_CHECK(hipFree(dQ)); tt[7] = now();
    if (sn::tuning().schur_profile)
        fprintf(stderr, "[api] Hessenberg n=%d: alloc+clear %.3f, upload A %.3f, upload Q %.3f, reduction %.3f, download A %.3f, download Q %.3f, free %.3f s\n",
            n, tt[1] - tt[0], tt[2] - tt[1], tt[3] - tt[2], tt[4] - tt[3], tt[5] - tt[4], tt[6] - tt[5], tt[7] - tt[6]);
    return rc == 0 ? STARNEIG_SUCCESS : STARNEIG_GENERIC_ERROR;
}

SN_API starneig_error_t starneig_SEP_SM_Hessenberg(
    int n, double A[], int ldA, double Q[], int ldQ)
{
    if (n < 1)      return -1;           // hessenberg/interface.c:175-179
    if (A == NULL)  return -2;
    if (ldA < n)    return -3;
    if (Q == NULL)  return -4;
    if (ldQ < n)    return -5;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    return starneig_SEP_SM_Hessenberg_expert(NULL, n, 0, n, A, ldA, Q, ldQ);
}

// Expert configuration -> SchurParams with the reference's range checks, in the reference's
// order: thresholds first (schur/core.c:2360-2384 -> STARNEIG_INVALID_CONFIGURATION), then
// schur/process_args.c:271-437 (-> STARNEIG_INVALID_ARGUMENTS).  Fields that describe the
// reference's tile/task organisation (tile_size, window_size, update_width/height,
// shift_origin -- unused by the reference as well) are range-checked exactly like there and
// then have no effect: H and Q are untiled here and the bulge window is fixed by the LDS of
// a CU.  shifts_per_window caps the bulges per chain; aed_parallel_hard_limit selects the
// host kernel below / the blocked device kernel above it (process_args.c:372-398).
static int schur_params_from_conf(struct starneig_schur_conf const *conf, sn::SchurParams &p,
    bool generalized)
{
    if (conf == NULL) return STARNEIG_SUCCESS;
    auto bad_thres = [](double t, bool lapack_ok) {
        return !(t == -1.0 || t == -2.0 || (lapack_ok && t == -3.0) || t > 0.0);
    };
    if (bad_thres(conf->left_threshold, true) || bad_thres(conf->right_threshold, true) ||
        bad_thres(conf->inf_threshold, false))
        return STARNEIG_INVALID_CONFIGURATION;
    if (conf->iteration_limit != -1 && conf->iteration_limit <= 0) return STARNEIG_INVALID_ARGUMENTS;
    if (conf->small_limit != -1 && conf->small_limit <= 2) return STARNEIG_INVALID_ARGUMENTS;
    if (conf->aed_window_size == -1 && conf->shift_count != -1 && conf->shift_count < 2)
        return STARNEIG_INVALID_ARGUMENTS;
    if (conf->aed_window_size != -1 && conf->shift_count == -1 && conf->aed_window_size <= 4)
        return STARNEIG_INVALID_ARGUMENTS;
    if (conf->aed_window_size != -1 && conf->shift_count != -1 &&
        conf->shift_count > conf->aed_window_size)
        return STARNEIG_INVALID_ARGUMENTS;
    if (conf->aed_nibble != -1 && (conf->aed_nibble <= 0 || conf->aed_nibble >= 100))
        return STARNEIG_INVALID_ARGUMENTS;
    if (conf->aed_parallel_soft_limit != -1 && conf->aed_parallel_soft_limit <= 0)
        return STARNEIG_INVALID_ARGUMENTS;
    if (conf->aed_parallel_hard_limit != -1 && conf->aed_parallel_hard_limit <= 0)
        return STARNEIG_INVALID_ARGUMENTS;
    if (conf->window_size != -1 && conf->window_size != -2 && conf->window_size < 5)
        return STARNEIG_INVALID_ARGUMENTS;
    if (conf->shifts_per_window != -1 && conf->shifts_per_window < 2)
        return STARNEIG_INVALID_ARGUMENTS;
    // (update_width / update_height <= 0: the reference warns and uses its default, :440-500)
    p.iteration_limit = conf->iteration_limit;
    p.small_limit = conf->small_limit;
    p.aed_nibble = conf->aed_nibble;
    // process_args.c:294-352: a lone shift count implies a window of twice that size, a lone
    // window implies half as many shifts, both: shifts <= 0.9 window
    if (conf->aed_window_size == -1 && conf->shift_count != -1) {
        p.aed_window_size = 2 * conf->shift_count; p.shift_count = conf->shift_count;
    } else if (conf->aed_window_size != -1 && conf->shift_count == -1) {
        p.aed_window_size = conf->aed_window_size; p.shift_count = conf->aed_window_size / 2;
    } else if (conf->aed_window_size != -1) {
        p.aed_window_size = conf->aed_window_size;
        p.shift_count = std::min(9 * conf->aed_window_size / 10, conf->shift_count);
    }
    p.shifts_per_window = conf->shifts_per_window;
    p.aed_parallel_hard_limit = conf->aed_parallel_hard_limit;
    p.threshold = conf->left_threshold;
    p.threshold_b = conf->right_threshold;
    p.threshold_inf = conf->inf_threshold;
    (void)generalized;
    return STARNEIG_SUCCESS;
}

SN_API starneig_error_t starneig_SEP_SM_Schur_expert(
    struct starneig_schur_conf *conf, int n, double H[], int ldH,
    double Q[], int ldQ, double real[], double imag[])
{
    if (n < 1)      return -2;           // schur/interface.c:198-202
    if (H == NULL)  return -3;
    if (ldH < n)    return -4;
    if (Q == NULL)  return -5;
    if (ldQ < n)    return -6;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    sn::SchurParams prm;
    int rc = schur_params_from_conf(conf, prm, false);
    if (rc != STARNEIG_SUCCESS) return rc;
    prm.host_threads = g_node.cores;

    if (!fits_device(n, 2, "starneig_SEP_SM_Schur")) return STARNEIG_GENERIC_ERROR;
    if (g_node.gpus > 1 && n >= 256 * g_node.gpus) {
        if (real == NULL || imag == NULL) real = imag = nullptr;
        return sn::node_team_schur(n, H, ldH, Q, ldQ, real, imag, prm, g_node.cores);
    }

    int const ld = (int)sn::roundup(n, 16);
    size_t const bytes = (size_t)ld * n * sizeof(double);
    double *dH = nullptr, *dQ = nullptr;
    SN_HIP_CHECK(hipMalloc((void **)&dH, bytes));
    SN_HIP_CHECK(hipMalloc((void **)&dQ, bytes));
    SN_HIP_CHECK(hipMemset(dH, 0, bytes));
    SN_HIP_CHECK(hipMemset(dQ, 0, bytes));
    to_device(dH, ld, H, ldH, n);
    to_device(dQ, ld, Q, ldQ, n);
    std::vector<double> wr, wi;
    double *pr = real, *pi = imag;
    if (real == NULL || imag == NULL) { pr = pi = nullptr; }    // schur/core.c:2501
    rc = sn::schur_device(nullptr, n, dH, ld, dQ, ld, pr, pi, prm, nullptr);
    SN_HIP_CHECK(hipStreamSynchronize(nullptr));
    to_host(H, ldH, dH, ld, n);
    to_host(Q, ldQ, dQ, ld, n);
    SN_HIP_CHECK(hipFree(dH));
    SN_HIP_CHECK(hipFree(dQ));
    return rc;
}

SN_API starneig_error_t starneig_SEP_SM_Schur(
    int n, double H[], int ldH, double Q[], int ldQ, double real[], double imag[])
{
    if (n < 1)      return -1;           // schur/interface.c:228-232
    if (H == NULL)  return -2;
    if (ldH < n)    return -3;
    if (Q == NULL)  return -4;
    if (ldQ < n)    return -5;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    return starneig_SEP_SM_Schur_expert(NULL, n, H, ldH, Q, ldQ, real, imag);
}

// common/helpers.c:47-101: evaluate a predicate on the eigenvalues of a Schur form
SN_API starneig_error_t starneig_SEP_SM_Select(
    int n, double S[], int ldS,
    int (*predicate)(double real, double imag, void *arg), void *arg,
    int selected[], int *num_selected)
{
    if (n < 1)              return -1;
    if (S == NULL)          return -2;
    if (ldS < n)            return -3;
    if (predicate == NULL)  return -4;
    if (selected == NULL)   return -6;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    std::vector<double> wr(n), wi(n);
    sn::host::extract_eigenvalues(n, S, ldS, wr.data(), wi.data());
    int count = 0;
    for (int i = 0; i < n; i++) {
        // a 2x2 block (S(i+1,i) != 0, also a non-standardised one with real eigenvalues) is
        // selected or rejected as a whole, by its first eigenvalue (common/helpers.c:70-93)
        if (i + 1 < n && S[(size_t)i * ldS + i + 1] != 0.0) {
            int sel = predicate(wr[i], wi[i], arg) ? 1 : 0;
            selected[i] = selected[i + 1] = sel;
            count += 2 * sel;
            i++;
        } else {
            selected[i] = predicate(wr[i], 0.0, arg) ? 1 : 0;
            count += selected[i];
        }
    }
    if (num_selected != NULL) *num_selected = count;
    return STARNEIG_SUCCESS;
}

// reorder/interface.c:190-203
SN_API void starneig_reorder_init_conf(struct starneig_reorder_conf *conf)
{
    conf->plan = STARNEIG_REORDER_DEFAULT_PLAN;
    conf->blueprint = STARNEIG_REORDER_DEFAULT_BLUEPRINT;
    conf->tile_size = STARNEIG_REORDER_DEFAULT_TILE_SIZE;
    conf->window_size = STARNEIG_REORDER_DEFAULT_WINDOW_SIZE;
    conf->values_per_chain = STARNEIG_REORDER_DEFAULT_VALUES_PER_CHAIN;
    conf->small_window_size = STARNEIG_REORDER_DEFAULT_SMALL_WINDOW_SIZE;
    conf->small_window_threshold = STARNEIG_REORDER_DEFAULT_SMALL_WINDOW_THRESHOLD;
    conf->update_width = STARNEIG_REORDER_DEFAULT_UPDATE_WIDTH;
    conf->update_height = STARNEIG_REORDER_DEFAULT_UPDATE_HEIGHT;
}

// reorder/core.c:470-651: plan / blueprint / window checks -> STARNEIG_INVALID_CONFIGURATION.
// Plans and blueprints describe the reference's task-insertion strategies; there is one schedule
// here, so valid values are accepted and have no effect.  window_size above 128 rows is clamped
// (the in-place update tiles own a whole window); values_per_chain caps the rows of selected
// blocks that travel together.
static int reorder_params_from_conf(struct starneig_reorder_conf const *conf, int &window, int &vpc)
{
    window = -1; vpc = -1;
    if (conf == NULL) return STARNEIG_SUCCESS;
    if (conf->plan < STARNEIG_REORDER_DEFAULT_PLAN || conf->plan > STARNEIG_REORDER_MULTI_PART_PLAN)
        return STARNEIG_INVALID_CONFIGURATION;
    if (conf->blueprint < STARNEIG_REORDER_DEFAULT_BLUEPRINT || conf->blueprint > STARNEIG_REORDER_CHAIN_INSERT_F)
        return STARNEIG_INVALID_CONFIGURATION;
    if (conf->tile_size != STARNEIG_REORDER_DEFAULT_TILE_SIZE && conf->tile_size < 8)
        return STARNEIG_INVALID_CONFIGURATION;
    if (conf->window_size != STARNEIG_REORDER_DEFAULT_WINDOW_SIZE &&
        conf->window_size != STARNEIG_REORDER_ROUNDED_WINDOW_SIZE && conf->window_size < 4)
        return STARNEIG_INVALID_CONFIGURATION;
    if (conf->values_per_chain != STARNEIG_REORDER_DEFAULT_VALUES_PER_CHAIN && conf->values_per_chain < 1)
        return STARNEIG_INVALID_CONFIGURATION;
    if (conf->small_window_size != STARNEIG_REORDER_DEFAULT_SMALL_WINDOW_SIZE && conf->small_window_size < 4)
        return STARNEIG_INVALID_CONFIGURATION;
    if (conf->window_size > 0) window = conf->window_size;
    if (conf->values_per_chain > 0) vpc = conf->values_per_chain;
    return STARNEIG_SUCCESS;
}

SN_API starneig_error_t starneig_SEP_SM_ReorderSchur_expert(
    struct starneig_reorder_conf *conf, int n, int selected[],
    double S[], int ldS, double Q[], int ldQ, double real[], double imag[])
{
    if (n < 1)              return -2;       // reorder/interface.c:213-218
    if (selected == NULL)   return -3;
    if (S == NULL)          return -4;
    if (ldS < n)            return -5;
    if (Q == NULL)          return -6;
    if (ldQ < n)            return -7;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    int window, vpc;
    int rc = reorder_params_from_conf(conf, window, vpc);
    if (rc != STARNEIG_SUCCESS) return rc;

    if (!fits_device(n, 2, "starneig_SEP_SM_ReorderSchur")) return STARNEIG_GENERIC_ERROR;
    int const ld = (int)sn::roundup(n, 16);
    size_t const bytes = (size_t)ld * n * sizeof(double);
    double *dS = nullptr, *dQ = nullptr;
    SN_HIP_CHECK(hipMalloc((void **)&dS, bytes));
    SN_HIP_CHECK(hipMalloc((void **)&dQ, bytes));
    SN_HIP_CHECK(hipMemset(dS, 0, bytes));
    SN_HIP_CHECK(hipMemset(dQ, 0, bytes));
    to_device(dS, ld, S, ldS, n);
    to_device(dQ, ld, Q, ldQ, n);
    if (real == NULL || imag == NULL) real = imag = nullptr;
    rc = sn::reorder_schur_device(nullptr, n, selected, dS, ld, dQ, ld, real, imag, window, vpc, nullptr, g_node.cores);
    SN_HIP_CHECK(hipStreamSynchronize(nullptr));
    to_host(S, ldS, dS, ld, n);
    to_host(Q, ldQ, dQ, ld, n);
    SN_HIP_CHECK(hipFree(dS));
    SN_HIP_CHECK(hipFree(dQ));
    return rc;
}

SN_API starneig_error_t starneig_SEP_SM_ReorderSchur(
    int n, int selected[], double S[], int ldS, double Q[], int ldQ, double real[], double imag[])
{
    if (n < 1)              return -1;       // reorder/interface.c:244-249
    if (selected == NULL)   return -2;
    if (S == NULL)          return -3;
    if (ldS < n)            return -4;
    if (Q == NULL)          return -5;
    if (ldQ < n)            return -6;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    return starneig_SEP_SM_ReorderSchur_expert(NULL, n, selected, S, ldS, Q, ldQ, real, imag);
}

SN_API starneig_error_t starneig_amd_reorder_schur_device(
    int n, int *selected, double *dS, int ldS, double *dQ, int ldQ, double *real, double *imag,
    struct starneig_reorder_conf *conf, void *stream, double *stats)
{
    if (n < 1)                 return -1;
    if (selected == NULL)      return -2;
    if (dS == NULL)            return -3;
    if (ldS < n)               return -4;
    if (dQ != NULL && ldQ < n) return -6;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    int window, vpc;
    int rc = reorder_params_from_conf(conf, window, vpc);
    if (rc != STARNEIG_SUCCESS) return rc;
    if (real == NULL || imag == NULL) real = imag = nullptr;
    return sn::reorder_schur_device((hipStream_t)stream, n, selected, dS, ldS, dQ, ldQ, real, imag,
        window, vpc, stats, g_node.cores);
}

// common/combined.c:46-98
SN_API starneig_error_t starneig_SEP_SM_Reduce(
    int n, double A[], int ldA, double Q[], int ldQ, double real[], double imag[],
    int (*predicate)(double real, double imag, void *arg), void *arg,
    int selected[], int *num_selected)
{
    if (n < 1)      return -1;
    if (A == NULL)  return -2;
    if (ldA < n)    return -3;
    if (Q == NULL)  return -4;
    if (ldQ < n)    return -5;           // common/combined.c:57-61: nothing beyond -5
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    starneig_error_t rc = starneig_SEP_SM_Hessenberg(n, A, ldA, Q, ldQ);
    if (rc != STARNEIG_SUCCESS) return rc;
    rc = starneig_SEP_SM_Schur(n, A, ldA, Q, ldQ, real, imag);
    if (rc != STARNEIG_SUCCESS) return rc;
    if (predicate != NULL) {
        std::vector<int> own;
        if (selected == NULL) { own.resize(n); selected = own.data(); }
        rc = starneig_SEP_SM_Select(n, A, ldA, predicate, arg, selected, num_selected);
        if (rc != STARNEIG_SUCCESS) return rc;
        rc = starneig_SEP_SM_ReorderSchur(n, selected, A, ldA, Q, ldQ, real, imag);
    }
    return rc;
}

// ---- generalized Schur (gep_sm.h) ---------------------------------------------
SN_API starneig_error_t starneig_GEP_SM_Schur_expert(
    struct starneig_schur_conf *conf, int n, double H[], int ldH, double R[], int ldR,
    double Q[], int ldQ, double Z[], int ldZ, double real[], double imag[], double beta[])
{
    if (n < 1)      return -2;
    if (H == NULL)  return -3;
    if (ldH < n)    return -4;
    if (R == NULL)  return -5;
    if (ldR < n)    return -6;
    if (Q == NULL)  return -7;
    if (ldQ < n)    return -8;
    if (Z == NULL)  return -9;
    if (ldZ < n)    return -10;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    sn::SchurParams prm;
    int rc = schur_params_from_conf(conf, prm, true);
    if (rc != STARNEIG_SUCCESS) return rc;
    prm.host_threads = g_node.cores;

    int const ld = (int)sn::roundup(n, 16);
    size_t const bytes = (size_t)ld * n * sizeof(double);
    if (!fits_device(n, 4, "starneig_GEP_SM_Schur")) return STARNEIG_GENERIC_ERROR;
    double *host[4] = {H, R, Q, Z};
    int const lds[4] = {ldH, ldR, ldQ, ldZ};
    double *dev[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 4; i++) {
        SN_HIP_CHECK(hipMalloc((void **)&dev[i], bytes));
        SN_HIP_CHECK(hipMemset(dev[i], 0, bytes));
        to_device(dev[i], ld, host[i], lds[i], n);
    }
    if (real == NULL || imag == NULL || beta == NULL) real = imag = beta = nullptr;
    rc = sn::gep_schur_device(nullptr, n, dev[0], ld, dev[1], ld, dev[2], ld, dev[3], ld,
        real, imag, beta, prm, nullptr);
    SN_HIP_CHECK(hipStreamSynchronize(nullptr));
    for (int i = 0; i < 4; i++) {
        to_host(host[i], lds[i], dev[i], ld, n);
        SN_HIP_CHECK(hipFree(dev[i]));
    }
    return rc;
}

SN_API starneig_error_t starneig_GEP_SM_Schur(
    int n, double H[], int ldH, double R[], int ldR, double Q[], int ldQ, double Z[], int ldZ,
    double real[], double imag[], double beta[])
{
    if (n < 1)      return -1;
    if (H == NULL)  return -2;
    if (ldH < n)    return -3;
    if (R == NULL)  return -4;
    if (ldR < n)    return -5;
    if (Q == NULL)  return -6;
    if (ldQ < n)    return -7;
    if (Z == NULL)  return -8;
    if (ldZ < n)    return -9;           // schur/interface.c:286-294: nothing beyond -9
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    return starneig_GEP_SM_Schur_expert(NULL, n, H, ldH, R, ldR, Q, ldQ, Z, ldZ, real, imag, beta);
}

// ---- Hessenberg-triangular reduction (gep_sm.h; wrappers/lapack.c:45-176) --------------
SN_API starneig_error_t starneig_GEP_SM_HessenbergTriangular(
    int n, double A[], int ldA, double B[], int ldB, double Q[], int ldQ, double Z[], int ldZ)
{
    if (n < 1)      return -1;
    if (A == NULL)  return -2;
    if (ldA < n)    return -3;
    if (B == NULL)  return -4;
    if (ldB < n)    return -5;
    if (Q == NULL)  return -6;
    if (ldQ < n)    return -7;
    if (Z == NULL)  return -8;
    if (ldZ < n)    return -9;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;

    int const ld = (int)sn::roundup(n, 16);
    size_t const bytes = (size_t)ld * n * sizeof(double);
    if (!fits_device(n, 4, "starneig_GEP_SM_HessenbergTriangular")) return STARNEIG_GENERIC_ERROR;
    double *host[4] = {A, B, Q, Z};
    int const lds[4] = {ldA, ldB, ldQ, ldZ};
    double *dev[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 4; i++) {
        SN_HIP_CHECK(hipMalloc((void **)&dev[i], bytes));
        SN_HIP_CHECK(hipMemset(dev[i], 0, bytes));
        to_device(dev[i], ld, host[i], lds[i], n);
    }
    int const rc = sn::hessenberg_triangular_device(nullptr, n, dev[0], ld, dev[1], ld, dev[2], ld, dev[3], ld, nullptr);
    SN_HIP_CHECK(hipStreamSynchronize(nullptr));
    for (int i = 0; i < 4; i++) {
        to_host(host[i], lds[i], dev[i], ld, n);
        SN_HIP_CHECK(hipFree(dev[i]));
    }
    return rc;
}

// common/combined.c:98-153: HessenbergTriangular + Schur (+ Select + ReorderSchur with a predicate;
// the generalized reordering is not on this path, a predicate is refused)
SN_API starneig_error_t starneig_GEP_SM_Reduce(
    int n, double A[], int ldA, double B[], int ldB, double Q[], int ldQ, double Z[], int ldZ,
    double real[], double imag[], double beta[],
    int (*predicate)(double real, double imag, double beta, void *arg), void *arg,
    int selected[], int *num_selected)
{
    if (n < 1)      return -1;
    if (A == NULL)  return -2;
    if (ldA < n)    return -3;
    if (B == NULL)  return -4;
    if (ldB < n)    return -5;
    if (Q == NULL)  return -6;
    if (ldQ < n)    return -7;
    if (Z == NULL)  return -8;
    if (ldZ < n)    return -9;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    if (predicate) {
        (void)arg; (void)selected; (void)num_selected;
        fprintf(stderr, "[starneig-amd] starneig_GEP_SM_Reduce: the generalized ReorderSchur step is not "
            "part of this library; call without a predicate.\n");
        return STARNEIG_GENERIC_ERROR;
    }
    int rc = starneig_GEP_SM_HessenbergTriangular(n, A, ldA, B, ldB, Q, ldQ, Z, ldZ);
    if (rc != STARNEIG_SUCCESS) return rc;
    return starneig_GEP_SM_Schur(n, A, ldA, B, ldB, Q, ldQ, Z, ldZ, real, imag, beta);
}

SN_API starneig_error_t starneig_amd_hessenberg_triangular_device(
    int n, double *dA, int ldA, double *dB, int ldB, double *dQ, int ldQ, double *dZ, int ldZ,
    void *stream, double *stats)
{
    if (n < 1)                 return -1;
    if (dA == NULL)            return -2;
    if (ldA < n)               return -3;
    if (dB == NULL)            return -4;
    if (ldB < n)               return -5;
    if (dQ != NULL && ldQ < n) return -7;
    if (dZ != NULL && ldZ < n) return -9;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    return sn::hessenberg_triangular_device((hipStream_t)stream, n, dA, ldA, dB, ldB, dQ, ldQ, dZ, ldZ, stats);
}

// ---- device-pointer extension -------------------------------------------------

SN_API starneig_error_t starneig_amd_gep_schur_device(
    int n, double *dH, int ldH, double *dR, int ldR, double *dQ, int ldQ, double *dZ, int ldZ,
    double *real, double *imag, double *beta,
    struct starneig_schur_conf *conf, void *stream, double *stats)
{
    if (n < 1)                 return -1;
    if (dH == NULL)            return -2;
    if (ldH < n)               return -3;
    if (dR == NULL)            return -4;
    if (ldR < n)               return -5;
    if (dQ != NULL && ldQ < n) return -7;
    if (dZ != NULL && ldZ < n) return -9;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    sn::SchurParams prm;
    int rc = schur_params_from_conf(conf, prm, true);
    if (rc != STARNEIG_SUCCESS) return rc;
    prm.host_threads = g_node.cores;
    sn::SchurStats st;
    hipStream_t s = (hipStream_t)stream;
    if (real == NULL || imag == NULL || beta == NULL) real = imag = beta = nullptr;
    rc = sn::gep_schur_device(s, n, dH, ldH, dR, ldR, dQ, ldQ, dZ, ldZ, real, imag, beta, prm, &st);
    SN_HIP_CHECK(hipStreamSynchronize(s));
    if (stats) {
        stats[0] = st.total_ms; stats[1] = st.sweeps; stats[2] = st.aeds;
        stats[3] = st.small_solves; stats[4] = st.chase_launches; stats[5] = st.gemm_flops;
        stats[6] = st.aed_host_s; stats[7] = st.wait_s;
    }
    return rc;
}

SN_API starneig_error_t starneig_amd_schur_device(
    int n, double *dH, int ldH, double *dQ, int ldQ, double *real, double *imag,
    struct starneig_schur_conf *conf, void *stream, double *stats)
{
    if (n < 1)                 return -1;
    if (dH == NULL)            return -2;
    if (ldH < n)               return -3;
    if (dQ != NULL && ldQ < n) return -5;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    sn::SchurParams prm;
    int rc = schur_params_from_conf(conf, prm, false);
    if (rc != STARNEIG_SUCCESS) return rc;
    prm.host_threads = g_node.cores;
    sn::SchurStats st;
    hipStream_t s = (hipStream_t)stream;
    if (real == NULL || imag == NULL) real = imag = nullptr;
    rc = sn::schur_device(s, n, dH, ldH, dQ, ldQ, real, imag, prm, &st);
    SN_HIP_CHECK(hipStreamSynchronize(s));
    if (stats) {
        stats[0] = st.total_ms; stats[1] = st.sweeps; stats[2] = st.aeds;
        stats[3] = st.small_solves; stats[4] = st.chase_launches; stats[5] = st.gemm_flops;
        stats[6] = st.aed_host_s; stats[7] = st.wait_s;
    }
    return rc;
}

static starneig_error_t schur_replica(
    int n, double *dH, int ldH, double *dQrows, int ldQ, int q_rows, int rank, int world, double *real, double *imag,
    struct starneig_schur_conf *conf, void *stream, double *stats);

SN_API starneig_error_t starneig_amd_schur_rows_device(
    int n, double *dH, int ldH, double *dQrows, int ldQ, int q_rows, double *real, double *imag,
    struct starneig_schur_conf *conf, void *stream, double *stats)
{
    return schur_replica(n, dH, ldH, dQrows, ldQ, q_rows, 0, 1, real, imag, conf, stream, stats);
}

SN_API starneig_error_t starneig_amd_schur_sharded_device(
    int n, double *dH, int ldH, double *dQrows, int ldQ, int q_rows, int rank, int world, double *real, double *imag,
    struct starneig_schur_conf *conf, void *stream, double *stats)
{
    if (world < 1 || rank < 0 || rank >= world) return -7;
    return schur_replica(n, dH, ldH, dQrows, ldQ, q_rows, rank, world, real, imag, conf, stream, stats);
}

static starneig_error_t schur_replica(
    int n, double *dH, int ldH, double *dQrows, int ldQ, int q_rows, int rank, int world, double *real, double *imag,
    struct starneig_schur_conf *conf, void *stream, double *stats)
{
    if (n < 1)                 return -1;
    if (dH == NULL)            return -2;
    if (ldH < n)               return -3;
    if (dQrows == NULL)        return -4;
    if (ldQ < q_rows)          return -5;
    if (q_rows < 0 || q_rows > n) return -6;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    sn::SchurParams prm;
    int rc = schur_params_from_conf(conf, prm, false);
    if (rc != STARNEIG_SUCCESS) return rc;
    prm.host_threads = g_node.cores;
    sn::SchurStats st;
    hipStream_t s = (hipStream_t)stream;
    if (real == NULL || imag == NULL) real = imag = nullptr;
    prm.shard_rank = rank; prm.shard_world = world;
    rc = sn::schur_device(s, n, dH, ldH, dQrows, ldQ, real, imag, prm, &st, q_rows);
    SN_HIP_CHECK(hipStreamSynchronize(s));
    if (stats) {
        stats[0] = st.total_ms; stats[1] = st.sweeps; stats[2] = st.aeds;
        stats[3] = st.small_solves; stats[4] = st.chase_launches; stats[5] = st.gemm_flops;
        stats[6] = st.aed_host_s; stats[7] = st.wait_s;
    }
    return rc;
}

SN_API starneig_error_t starneig_amd_hessenberg_device(
    int n, int begin, int end, int panel_width,
    double *dA, int ldA, double *dQ, int ldQ, void *stream, double *stats)
{
    if (n < 1)                 return -1;
    if (begin < 0)             return -2;
    if (n < end)               return -3;
    if (dA == NULL)            return -5;
    if (ldA < n)               return -6;
    if (dQ != NULL && ldQ < n) return -8;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    if (panel_width <= 0) panel_width = default_panel_width(n);
    if (panel_width < 8) return STARNEIG_INVALID_CONFIGURATION;
    sn::HessenbergTimings tm;
    if (stats) {   // in: [7] = k, an integer in [1, 2^20]; anything else is "off"
        double const k_in = stats[7];
        tm.sample_every = (k_in >= 1.0 && k_in <= 1048576.0 && k_in == (double)(long)k_in) ? (int)k_in : 0;
    }
    hipStream_t s = (hipStream_t)stream;
    int rc = sn::hessenberg_device(s, n, begin, end, panel_width, dA, ldA, dQ, ldQ,
        stats ? &tm : nullptr);
    SN_HIP_CHECK(hipStreamSynchronize(s));
    if (stats) {
        stats[0] = tm.total_ms; stats[1] = tm.gemv_bytes; stats[2] = tm.gemm_flops;
        stats[3] = tm.sampled_ms; stats[4] = tm.sampled_bytes;
        stats[5] = (double)tm.gemv_launches; stats[6] = (double)tm.sampled_launches;
        stats[8] = tm.gemm_ms_main; stats[9] = tm.gemm_flops_main; stats[10] = tm.gemm_ms_side;
        stats[11] = tm.gemm_ms_fused; stats[12] = tm.gemm_flops_fused;
    }
    return rc == 0 ? STARNEIG_SUCCESS : STARNEIG_GENERIC_ERROR;
}

SN_API int starneig_amd_hessenberg_panel_ld(int n, int panel_width)
{
    if (panel_width <= 0) panel_width = default_panel_width(n);
    return sn::hessenberg_panel_ld(n, panel_width);
}

namespace {
struct NativeComm { double *buf[5]; hipStream_t s; };        // buffer ids of sn::HessComm: y, panel, W, A, Q
void native_allreduce(void *ctx, int buffer, long offset, long count)
{
    NativeComm *c = (NativeComm *)ctx;
    if (sn::rccl_allreduce_sum(c->buf[buffer] + offset, count, c->s) != 0) abort();   // a rank that drops out would hang the others
}
void native_broadcast(void *ctx, int buffer, long offset, long count, int root)
{
    NativeComm *c = (NativeComm *)ctx;
    if (sn::rccl_broadcast(c->buf[buffer] + offset, count, root, c->s) != 0) abort();
}
}

SN_API int starneig_amd_rccl_unique_id(void *id128) { return sn::rccl_unique_id(id128); }
SN_API int starneig_amd_rccl_init(int rank, int world, void const *id128) { return sn::rccl_init(rank, world, id128); }
SN_API void starneig_amd_rccl_finalize(void) { sn::rccl_finalize(); }
SN_API int starneig_amd_rccl_allreduce_sum(double *dbuf, long count, void *stream) { return sn::rccl_allreduce_sum(dbuf, count, (hipStream_t)stream); }
SN_API int starneig_amd_rccl_broadcast(double *dbuf, long count, int root, void *stream) { return sn::rccl_broadcast(dbuf, count, root, (hipStream_t)stream); }

SN_API starneig_error_t starneig_amd_hessenberg_sharded_device(
    int n, int panel_width, double *dA, int ldA, double *dQ, int ldQ,
    double *dY, double *dP, double *dW, long w_capacity,
    int rank, int world,
    void (*allreduce_sum)(void *ctx, int buffer, long offset, long count),
    void (*broadcast)(void *ctx, int buffer, long offset, long count, int root),
    void *ctx, void *stream, double *stats)
{
    if (n < 1)                 return -1;
    if (dA == NULL)            return -3;
    if (ldA < n)               return -4;
    if (dQ != NULL && ldQ < n) return -6;
    if (dY == NULL || dP == NULL || dW == NULL) return STARNEIG_INVALID_ARGUMENTS;
    if (world < 1 || rank < 0 || rank >= world) return STARNEIG_INVALID_ARGUMENTS;
    // both callbacks NULL: the collectives go to RCCL directly, on `stream` (starneig_amd_rccl_init first)
    bool const native = !allreduce_sum && !broadcast;
    if (!native && (!allreduce_sum || !broadcast)) return STARNEIG_INVALID_ARGUMENTS;
    if (native && !sn::rccl_ready(rank, world)) return STARNEIG_INVALID_ARGUMENTS;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    if (panel_width <= 0) panel_width = default_panel_width(n);
    if (panel_width < 8) return STARNEIG_INVALID_CONFIGURATION;
    hipStream_t s = (hipStream_t)stream;
    NativeComm nc{{dY, dP, dW, dA, dQ}, s};
    sn::HessComm comm{rank, world, native ? native_allreduce : allreduce_sum, native ? native_broadcast : broadcast,
                      native ? (void *)&nc : ctx};
    sn::HessenbergTimings tm;
    // `stats` holds 32 zero-initialised doubles; stats[7] = k on entry: every k-th gemv launch (with its
    // all-reduce) and every per-panel collective is event-timed (include/starneig_amd.h).  Only an integer in
    // [1, 2^20] switches the detailed report on -- an uninitialised array must not
    double const k_in = stats ? stats[7] : 0.0;
    bool const detailed = stats && k_in >= 1.0 && k_in <= 1048576.0 && k_in == (double)(long)k_in;
    if (detailed) tm.sample_every = (int)k_in;
    int rc = sn::hessenberg_sharded_device(s, n, panel_width, dA, ldA, dQ, ldQ, dY, dP, dW,
        w_capacity, comm, stats ? &tm : nullptr);
    SN_HIP_CHECK(hipStreamSynchronize(s));
    if (stats) { stats[0] = tm.total_ms; stats[1] = tm.gemv_bytes; stats[2] = tm.gemm_flops;
                 stats[5] = (double)tm.gemv_launches; }
    if (detailed) {
        stats[8] = tm.sampled_ms; stats[9] = tm.sampled_bytes; stats[10] = (double)tm.sampled_launches;
        stats[11] = (double)tm.allreduce_y_calls;
        for (int k = 0; k < 4; k++) {
            stats[12 + 3 * k] = tm.comm_ms[k]; stats[13 + 3 * k] = tm.comm_bytes[k]; stats[14 + 3 * k] = (double)tm.comm_calls[k];
        }
        stats[24] = sn::rccl_comm_count();
    }
    return rc == 0 ? STARNEIG_SUCCESS : STARNEIG_GENERIC_ERROR;
}

SN_API starneig_error_t starneig_amd_dgemm_device(
    char transA, char transB, int m, int n, int k, double alpha,
    double const *dA, int ldA, double const *dB, int ldB, double beta,
    double *dC, int ldC, void *stream)
{
    if (m < 0) return -3;
    if (n < 0) return -4;
    if (k < 0) return -5;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    sn::dgemm((hipStream_t)stream, transA, transB, m, n, k, alpha, dA, ldA, dB, ldB, beta, dC, ldC);
    return STARNEIG_SUCCESS;
}

SN_API starneig_error_t starneig_amd_lcg_fill_device(
    int m, int n, unsigned seed, int mode, double *dA, int ldA, void *stream)
{
    if (m < 0) return -1;
    if (n < 0) return -2;
    if (dA == NULL) return -5;
    if (ldA < m) return -6;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    sn::lcg_fill((hipStream_t)stream, m, n, seed, mode, dA, ldA);
    return STARNEIG_SUCCESS;
}

SN_API starneig_error_t starneig_amd_lcg_pencil_device(
    int n, unsigned seed, double *dH, int ldH, double *dR, int ldR, void *stream)
{
    if (n < 1) return -1;
    if (dH == NULL) return -3;
    if (ldH < n) return -4;
    if (dR == NULL) return -5;
    if (ldR < n) return -6;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    sn::lcg_pencil((hipStream_t)stream, n, seed, dH, ldH, dR, ldR);
    return STARNEIG_SUCCESS;
}

SN_API starneig_error_t starneig_amd_set_matrix_device(
    int m, int n, double value, double diag, double *dA, int ldA, void *stream)
{
    if (dA == NULL) return -5;
    if (ldA < m) return -6;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    sn::set_matrix((hipStream_t)stream, m, n, value, diag, dA, ldA);
    return STARNEIG_SUCCESS;
}

SN_API starneig_error_t starneig_amd_check_device(
    int n, double const *dQ, int ldQ, double const *dH, int ldH,
    double const *dA0, int ldA0, double *dWork1, double *dWork2,
    double out[3], void *stream)
{
    if (n < 1) return -1;
    if (!dQ || !dH || !dA0 || !dWork1 || !dWork2 || !out) return STARNEIG_INVALID_ARGUMENTS;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    hipStream_t s = (hipStream_t)stream;
    double *acc = nullptr;
    SN_HIP_CHECK(hipMalloc((void **)&acc, 4 * sizeof(double)));
    SN_HIP_CHECK(hipMemsetAsync(acc, 0, 4 * sizeof(double), s));
    // W1 = Q H ; W2 = W1 Q^T ; ||W2 - A||_F / ||A||_F       (checks.c:180-194)
    sn::dgemm_accurate(s, 'N', 'N', n, n, n, dQ, ldQ, dH, ldH, dWork1, n);
    sn::dgemm_accurate(s, 'N', 'T', n, n, n, dWork1, n, dQ, ldQ, dWork2, n);
    sn::sumsq_diff(s, n, n, dWork2, n, dA0, ldA0, 0.0, acc + 0);
    sn::sumsq_diff(s, n, n, dA0, ldA0, nullptr, 0, 0.0, acc + 1);
    // ||Q Q^T - I||_F / sqrt(n)                              (checks.c:196-208)
    sn::dgemm_accurate(s, 'N', 'T', n, n, n, dQ, ldQ, dQ, ldQ, dWork1, n);
    sn::sumsq_diff(s, n, n, dWork1, n, nullptr, 0, 1.0, acc + 2);
    sn::count_below(s, n, dH, ldH, acc + 3);
    double h[4];
    SN_HIP_CHECK(hipMemcpyAsync(h, acc, sizeof h, hipMemcpyDeviceToHost, s));
    SN_HIP_CHECK(hipStreamSynchronize(s));
    SN_HIP_CHECK(hipFree(acc));
    out[0] = std::ldexp(std::sqrt(h[0]) / std::sqrt(h[1]), 52);
    out[1] = std::ldexp(std::sqrt(h[2]) / std::sqrt((double)n), 52);
    out[2] = h[3];
    return STARNEIG_SUCCESS;
}

SN_API starneig_error_t starneig_amd_check_pencil_device(
    int n, double const *dQ, int ldQ, double const *dS, int ldS, double const *dZ, int ldZ,
    double const *dA0, int ldA0, double *dWork1, double *dWork2, double out[4], void *stream)
{
    if (n < 1) return -1;
    if (!dQ || !dS || !dZ || !dA0 || !dWork1 || !dWork2 || !out) return STARNEIG_INVALID_ARGUMENTS;
    if (!g_node.initialized) return STARNEIG_NOT_INITIALIZED;
    hipStream_t s = (hipStream_t)stream;
    double *acc = nullptr;
    SN_HIP_CHECK(hipMalloc((void **)&acc, 5 * sizeof(double)));
    SN_HIP_CHECK(hipMemsetAsync(acc, 0, 5 * sizeof(double), s));
    sn::dgemm_accurate(s, 'N', 'N', n, n, n, dQ, ldQ, dS, ldS, dWork1, n);
    sn::dgemm_accurate(s, 'N', 'T', n, n, n, dWork1, n, dZ, ldZ, dWork2, n);
    sn::sumsq_diff(s, n, n, dWork2, n, dA0, ldA0, 0.0, acc + 0);
    sn::sumsq_diff(s, n, n, dA0, ldA0, nullptr, 0, 0.0, acc + 1);
    sn::dgemm_accurate(s, 'N', 'T', n, n, n, dQ, ldQ, dQ, ldQ, dWork1, n);
    sn::sumsq_diff(s, n, n, dWork1, n, nullptr, 0, 1.0, acc + 2);
    sn::dgemm_accurate(s, 'N', 'T', n, n, n, dZ, ldZ, dZ, ldZ, dWork1, n);
    sn::sumsq_diff(s, n, n, dWork1, n, nullptr, 0, 1.0, acc + 3);
    sn::count_below(s, n, dS, ldS, acc + 4);
    double h[5];
    SN_HIP_CHECK(hipMemcpyAsync(h, acc, sizeof h, hipMemcpyDeviceToHost, s));
    SN_HIP_CHECK(hipStreamSynchronize(s));
    SN_HIP_CHECK(hipFree(acc));
    out[0] = std::ldexp(std::sqrt(h[0]) / std::sqrt(h[1]), 52);
    out[1] = std::ldexp(std::sqrt(h[2]) / std::sqrt((double)n), 52);
    out[2] = std::ldexp(std::sqrt(h[3]) / std::sqrt((double)n), 52);
    out[3] = h[4];
    return STARNEIG_SUCCESS;
}

} // extern "C"
