// Blocked Householder Hessenberg reduction on one MI355X.
//
// Rebuilds rows H0-H10 of SURVEY.md section 8a: the panel/column/update order of
// reference hessenberg/core.c:399-596 (+ delayed updates :301-349) with the
// arithmetic of hessenberg/cpu.c:50-560, as a static schedule of HIP kernels
// on one device-resident column-major matrix (no tiles, no task graph).
//
// Per panel column j (global pivot row piv = i+1+j) the dependent chain is three
// launches:
//   colA(j) : finish Y(:,j-1) = tau (y - Y w_v) from the gemv partials (cpu.c:253-270),
//             p' = P(:,j) - Y V(piv-1,:)^T (cpu.c:98-99), w = (V T)^T p' (cpu.c:109-120); the two
//             products with the columns of Y that exist before gemv(j-1) are formed in that launch's
//             shadow, not here
//   colC(j) : p'' = p' - V w (cpu.c:123-130), ||p''(piv+1:)||^2, V^T p''  (for w_v)
//   gemv(j) : y = A(i+1:end, piv:end) v  -- THE HBM-bound kernel (cpu.c:217-219,
//             cuda.cu:62-107): every trailing element is streamed once per column;
//             extra blocks of the same launch materialise V(:,j) and the new column
//             of VT = V*T (VT(:,j) = tau (v - VT w_v), which is V t_j + tau v with
//             t_j = -tau T w_v of cpu.c:277-284), in the gemv's shadow.
// T itself is never formed: every consumer needs V*T only.  Cross-workgroup sums
// (w, w_v, the norm) are ORDERED: every workgroup stores its partial vector, the workgroups
// b = s, s + 16, ... share slot s, and the one that draws the slot's last ticket adds the slot's
// partials up in workgroup order (slot_fold below); the consumers add the 16 slot sums in slot
// order.  Two reductions of one matrix give the same bits, and so do the replicas of a sharded
// reduction (rounds 1-5 used fp64 atomics into 8 slots: the reference's STARPU_COMMUTE
// accumulations, hessenberg/tasks.c:374,515,622, and like them different in the last bits from
// run to run).  The reflector scalars (LAPACK dlarfg, cpu.c:137-141) are recomputed by each consumer.
// All panel matrices (P,V,VT,Y) are indexed by GLOBAL row so that the 16-byte row
// pairs of the gemv stay aligned for every panel offset.
#include "common.h"
#include "tuning.h"
#include <vector>
#include <algorithm>
#include <cmath>

namespace sn {

typedef double d2 __attribute__((ext_vector_type(2)));

// ---- 16-lane (DPP row) all-reduce of a double --------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double x)
{
    x += dpp_f64<0x128>(x);   // row_ror:8
    x += dpp_f64<0x124>(x);   // row_ror:4
    x += dpp_f64<0x122>(x);   // row_ror:2
    x += dpp_f64<0x121>(x);   // row_ror:1
    return x;
}
__device__ __forceinline__ double wave_sum(double x)
{
    x = row16_sum(x);
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
}

constexpr int RB = 64;           // rows per workgroup in the row-parallel column kernels
constexpr int NG = 4;            // column groups (= waves) sharing one row in colA / finish
constexpr int CT = 64 * NG;      // threads of colA / finish
constexpr int NGC = 16;          // colC: its column loop is latency-bound, 16 groups cut it from 15 to 10 us
constexpr int CTC = 64 * NGC;    // (colA gets slower with 16: 23 -> 36 us)
constexpr int RBS = 32, NGS = 8;  // rows / column groups of a shadow block inside the gemv launch: with the
                                  // ~5 streaming workgroups per CU they must all be resident at once (8 per CU)
constexpr int GEMV_ROWS = 512;   // rows per workgroup of the big gemv (4 waves x 64 lanes x 2)
constexpr int MAX_SPLIT = 64;
constexpr int NSLOT = 16;        // slots of the ordered cross-workgroup sums (workgroup b belongs to slot b % NSLOT)
constexpr int MAXJ = 512;        // panel width limit of the column kernels

// slot sums (doubles): wsum[NSLOT][MAXJ], wvsum[NSLOT][MAXJ], nrm[NSLOT]; a slot without a workgroup keeps the
// zero of the panel's memset.  Every value is OVERWRITTEN by its slot's last arriver, never accumulated: no
// zeroing between columns, no second parity.
constexpr int ACC_WSUM = 0;
constexpr int ACC_WVSUM = NSLOT * MAXJ;
constexpr int ACC_NRM = 2 * NSLOT * MAXJ;
constexpr int ACC_TOTAL = ACC_NRM + NSLOT;

__device__ __forceinline__ double slot_sum(double const *__restrict__ base, int l)
{
    double s = 0.0;
    #pragma unroll
    for (int k = 0; k < NSLOT; k++) s += base[k * MAXJ + l];
    return s;
}

// LAPACK dlarfg scalars from alpha and the sum of squares below it (cpu.c:137-141)
__device__ __forceinline__ void reflector_scalars(double ssq, double alpha,
    double &scale, double &tau, double &beta)
{
    if (ssq == 0.0) { scale = 0.0; tau = 0.0; beta = alpha; return; }
    beta = -copysign(sqrt(alpha * alpha + ssq), alpha);
    tau = (beta - alpha) / beta;
    scale = 1.0 / (alpha - beta);
}

__device__ __forceinline__ double nrm_sum(double const *__restrict__ acc)
{
    double s = 0.0;
    #pragma unroll
    for (int k = 0; k < NSLOT; k++) s += acc[ACC_NRM + k];
    return s;
}

// y(g) = sum over the column splits of the gemv partials: 8 independent loads in flight (the
// plain loop waits for one L2 round trip per split -- 32 of them on the column chain)
__device__ __forceinline__ double split_sum(double const *__restrict__ ypart, int ldp, int g, int nsplit)
{
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int s = 0;
    for (; s + 8 <= nsplit; s += 8) {
        double x[8];
        #pragma unroll
        for (int q = 0; q < 8; q++) x[q] = ypart[(size_t)(s + q) * ldp + g];
        #pragma unroll
        for (int q = 0; q < 8; q++) a[q] += x[q];
    }
    for (; s < nsplit; s++) a[0] += ypart[(size_t)s * ldp + g];
    return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

// In-block transposed gemv: out[l] = sum_{r<RB} M[g0+r, l] * sp[r], l < ncols -- this workgroup's PARTIAL of
// the cross-workgroup sum (slot_fold adds the partials up).  CT threads = 16 row lanes x CT/16 column groups; 16
// lanes read 128 contiguous bytes of one column, the 16-lane DPP row reduces them, one write-through store per
// column (agent scope: the partial is read by another workgroup of this launch).
__device__ __forceinline__ void part_store(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int THREADS>
__device__ __forceinline__ void block_gemv_t_part(double const *__restrict__ M, int ldm,
    int g0, int ncols, double const *sp, double *__restrict__ out)
{
    int const rsub = threadIdx.x & 15, csub = threadIdx.x >> 4;
    constexpr int CG = THREADS / 16;
    double pr[RB / 16];
    #pragma unroll
    for (int it = 0; it < RB / 16; it++) pr[it] = sp[it * 16 + rsub];
    // four columns per trip: 16 independent loads in flight per lane (the loop is latency-bound)
    int l = csub;
    for (; l + 3 * CG < ncols; l += 4 * CG) {
        double x[4][RB / 16];
        #pragma unroll
        for (int q = 0; q < 4; q++) {
            double const *col = M + (size_t)(l + q * CG) * ldm + g0 + rsub;
            #pragma unroll
            for (int it = 0; it < RB / 16; it++) x[q][it] = col[it * 16];
        }
        #pragma unroll
        for (int q = 0; q < 4; q++) {
            double acc = 0.0;
            #pragma unroll
            for (int it = 0; it < RB / 16; it++) acc += x[q][it] * pr[it];
            acc = row16_sum(acc);
            if (rsub == 0) part_store(out + l + q * CG, acc);
        }
    }
    for (; l < ncols; l += CG) {
        double const *col = M + (size_t)l * ldm + g0 + rsub;
        double acc = 0.0;
        #pragma unroll
        for (int it = 0; it < RB / 16; it++) acc += col[it * 16] * pr[it];
        acc = row16_sum(acc);
        if (rsub == 0) part_store(out + l, acc);
    }
}

// The ordered cross-workgroup sum.  Every workgroup of the launch has stored its partial vector part[b][0:len)
// (+ an optional scalar spart[b]) WRITE-THROUGH (sc1: part_store); here every storing wave drains its stores, the
// workgroup meets at its barrier and one lane draws a ticket of slot b % NSLOT (relaxed, agent scope); the workgroup
// that draws the slot's LAST ticket adds the slot's partials up in workgroup order b = slot, slot + NSLOT, ... --
// EVERY load of a handed-off word an sc1 load, eight in flight, a fixed tree -- into sum[slot][0:len) (and
// ssum[slot]) and resets the counter.  (cdna_hip_programming.md section 6, Guideline 16, the counter form with
// write-through payload: no release fence on the producers, no acquire on the reducer -- with both fences in,
// the two column kernels took 3.7 us longer each, 0.15 s of a reduction at n = 20000; profiles/r6_column_chain.txt.)
// The NEXT launch reads the slot sums; a slot's value never depends on which workgroup came last.
template <int THREADS>
__device__ __forceinline__ void slot_fold(int nwg, int len, double const *__restrict__ part, double *__restrict__ sum,
    double const *__restrict__ spart, double *__restrict__ ssum, int *__restrict__ cnt)
{
    __shared__ int s_last;
    int const b = blockIdx.x, slot = b % NSLOT, members = (nwg - slot + NSLOT - 1) / NSLOT;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave: its write-through stores are out
    __syncthreads();
    if (threadIdx.x == 0) {
        int const last = __hip_atomic_fetch_add(cnt + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1;
        if (last) __hip_atomic_store(cnt + slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // (no instruction: the loads below stay below the ticket)
    auto ordered = [&](double const *p, size_t stride) {
        // member q of the slot goes to accumulator q % 8 (clamped loads, zeros beyond the last member: static
        // register indices), the eight accumulators meet in a fixed tree
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int q = 0; q < members; q += 8) {
            double x[8];
            #pragma unroll
            for (int u = 0; u < 8; u++)
                x[u] = __hip_atomic_load(p + (size_t)(slot + min(q + u, members - 1) * NSLOT) * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            #pragma unroll
            for (int u = 0; u < 8; u++) a[u] += (q + u < members) ? x[u] : 0.0;
        }
        return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    };
    for (int l = threadIdx.x; l < len; l += THREADS) sum[slot * MAXJ + l] = ordered(part + l, MAXJ);
    if (spart && threadIdx.x == THREADS - 1) ssum[slot] = ordered(spart, 1);
}

// ---- device-side exchange of the sharded gemv's result (HessExchange, common.h) ------------------------------
// Consumer side: the rows [g_lo, g_hi] of this workgroup lie in at most two row tiles of the gemv launch;
// 2 * world lanes poll the flags (system-scope relaxed loads; bounded: a rank that died must not hang the
// others), one acquire for the workgroup, then every row sums the world slots in rank order.
__device__ __forceinline__ void exchange_wait(HessExchange const &x, int seq, int tile_lo, int tile_hi)
{
    int const tid = threadIdx.x;
    if (tid < 2 * x.world) {
        int const r = tid >> 1, tile = (tid & 1) ? tile_hi : tile_lo;
        int const *f = x.flags[x.rank] + r * HESS_MAX_ROW_TILES + tile;
        // (bounded: ~1 s, and once a wait of this rank has timed out the reduction is lost anyway -- the rest of
        // it does not wait at all, the host reports the failure)
        long spins = 0;
        bool const lost = __hip_atomic_load(x.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
        while (!lost && __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq < 0) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1L << 22)) { __hip_atomic_store(x.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}
__device__ __forceinline__ double exchange_sum(HessExchange const &x, int seq, int ldp, int g)
{
    double const *base = x.slots[x.rank] + (size_t)(seq & 1) * x.world * ldp + g;
    double s = __hip_atomic_load(base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (int r = 1; r < x.world; r++)
        s += __hip_atomic_load(base + (size_t)r * ldp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return s;
}
// a rank that owns no column block right of the pivot publishes zeros (one workgroup per row tile)
__global__ __launch_bounds__(256)
void hess_exchange_zero_kernel(HessExchange x, int seq, int R0, int E, int ldp)
{
    int const tile = blockIdx.x, g = (R0 & ~15) + tile * 512 + threadIdx.x * 2;
    for (int r = 0; r < x.world; r++) {
        double *dst = x.slots[r] + ((size_t)(seq & 1) * x.world + x.rank) * ldp;
        if (g >= R0 && g < E) __hip_atomic_store(dst + g, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (g + 1 >= R0 && g + 1 < E) __hip_atomic_store(dst + g + 1, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int r = 0; r < x.world; r++)
            __hip_atomic_store(x.flags[r] + x.rank * HESS_MAX_ROW_TILES + tile, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// colA(j), j >= 1.  The two products with the OLD columns of Y that this step needs,
//   t1 = Y(:,0:j-1) w_v   (for Y(:,j-1) = tau (y - t1), cpu.c:267-270)  and
//   t2 = Y(:,0:j-1) V(piv-1,0:j-1)^T   (for p' = P(:,j) - Y V(piv-1,:)^T, cpu.c:98-99),
// depend on nothing that gemv(j-1) computes: its shadow blocks form them while the trailing matrix
// streams (a pass over the m x j panel factor Y that used to sit on the column chain).  What is left
// here: one round of loads, the new column of Y, p', and w = (V T)^T p'.
__global__ __launch_bounds__(CT)
void hess_colA_kernel(int R0, int E, int j, int ldp,
    double *__restrict__ P, double const *__restrict__ VT,
    double *__restrict__ Y, double const *__restrict__ ypart, int nsplit,
    double const *__restrict__ t12, double *__restrict__ acc, double const *__restrict__ scal,
    double *__restrict__ part, int *__restrict__ cnt,
    HessExchange x = HessExchange{}, int seq = 0)
{
    __shared__ double s_p[RB], s_scal[3];
    int const tid = threadIdx.x;
    int const r = tid & (RB - 1), h = tid >> 6;
    int const g0 = R0 + blockIdx.x * RB;
    int const g = g0 + r;
    int const pivprev = R0 + j - 1;
    // scalars of column j-1 as published by gemv(j-1): P(piv-1, j-1) itself is overwritten
    // with beta by one block of THIS launch, so it must not be re-read here
    if (tid < 3) s_scal[tid] = scal[4 * (j - 1) + tid];
    if (x.world) {
        int const base = R0 & ~15;
        exchange_wait(x, seq, (g0 - base) / GEMV_ROWS, (min(g0 + RB, E) - 1 - base) / GEMV_ROWS);
    }
    double ysum = 0.0, pj = 0.0, t1 = 0.0, t2 = 0.0;
    if (h == 0 && g < E) {
        ysum = x.world ? exchange_sum(x, seq, ldp, g) : split_sum(ypart, ldp, g, nsplit);
        pj = P[(size_t)j * ldp + g];
        t1 = t12[g]; t2 = t12[ldp + g];
    }
    __syncthreads();
    if (h == 0) {
        double pval = 0.0;
        if (g < E) {
            double const tau = s_scal[1], beta = s_scal[2];
            double const ynew = tau * (ysum - t1);             // cpu.c:267-270
            Y[(size_t)(j - 1) * ldp + g] = ynew;
            pval = pj - (t2 + ynew);                           // cpu.c:98-99; V(piv-1, j-1) = 1
            P[(size_t)j * ldp + g] = pval;
            // column j-1 of P is final: beta on the sub-diagonal, zeros below (cpu.c:153-154)
            if (g == pivprev) P[(size_t)(j - 1) * ldp + g] = beta;
            else if (g > pivprev) P[(size_t)(j - 1) * ldp + g] = 0.0;
        }
        s_p[r] = pval;
    }
    __syncthreads();
    // this workgroup's share of w = VT(rows,0:j)^T p'   (rows past E contribute 0 through s_p), then the ordered sum
    block_gemv_t_part<CT>(VT, ldp, g0, j, s_p, part + (size_t)blockIdx.x * MAXJ);
    slot_fold<CT>(gridDim.x, j, part, acc + ACC_WSUM, nullptr, nullptr, cnt);
}

// After the last column of a panel: finish Y(:,nb-1) and finalize P(:,nb-1).
__global__ __launch_bounds__(CT)
void hess_finish_kernel(int R0, int E, int j /* = nb */, int ldp,
    double *__restrict__ P, double const *__restrict__ V, double *__restrict__ Y,
    double const *__restrict__ ypart, int nsplit, double const *__restrict__ acc,
    double const *__restrict__ scal, HessExchange x = HessExchange{}, int seq = 0)
{
    __shared__ double s_wv[MAXJ], s_y[NG - 1][RB], s_scal[3];
    int const tid = threadIdx.x;
    int const r = tid & (RB - 1), h = tid >> 6;
    int const g = R0 + blockIdx.x * RB + r;
    int const pivprev = R0 + j - 1;
    if (tid < 3) s_scal[tid] = scal[4 * (j - 1) + tid];
    if (x.world) {
        int const base = R0 & ~15, g0 = R0 + blockIdx.x * RB;
        exchange_wait(x, seq, (g0 - base) / GEMV_ROWS, (min(g0 + RB, E) - 1 - base) / GEMV_ROWS);
    }
    __syncthreads();
    for (int l = tid; l < j - 1; l += CT)
        s_wv[l] = V[(size_t)l * ldp + pivprev]
            + s_scal[0] * slot_sum(acc + ACC_WVSUM, l);
    __syncthreads();
    double yacc = 0.0;
    if (g < E)
        for (int l = h; l < j - 1; l += NG) yacc += Y[(size_t)l * ldp + g] * s_wv[l];
    if (h > 0) s_y[h - 1][r] = yacc;
    __syncthreads();
    if (h == 0 && g < E) {
        double const tau = s_scal[1], beta = s_scal[2];
        #pragma unroll
        for (int q = 0; q < NG - 1; q++) yacc += s_y[q][r];
        double ysum = x.world ? exchange_sum(x, seq, ldp, g) : split_sum(ypart, ldp, g, nsplit);
        Y[(size_t)(j - 1) * ldp + g] = tau * (ysum - yacc);
        if (g == pivprev) P[(size_t)(j - 1) * ldp + g] = beta;
        else if (g > pivprev) P[(size_t)(j - 1) * ldp + g] = 0.0;
    }
}

// colC(j): p'' = p' - V(:,0:j) w ; norm^2 below the pivot ; V^T p''
__global__ __launch_bounds__(CTC)
void hess_colC_kernel(int R0, int E, int j, int ldp,
    double *__restrict__ P, double const *__restrict__ V, double *__restrict__ acc,
    double *__restrict__ part, double *__restrict__ npart, int *__restrict__ cnt)
{
    __shared__ double s_w[MAXJ], s_p[RB], s_t[NGC - 1][RB];
    int const tid = threadIdx.x;
    int const r = tid & (RB - 1), h = tid >> 6;
    int const g0 = R0 + blockIdx.x * RB;
    int const g = g0 + r;
    int const piv = R0 + j;
    for (int l = tid; l < j; l += CTC)
        s_w[l] = slot_sum(acc + ACC_WSUM, l);
    // requested in the same round as the slot sums (they depend on nothing computed here): this thread's
    // entry of column j and the first four entries of its row of V -- same arithmetic, one round trip less
    double const *vrow = V + g;
    double pj_early = 0.0, x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0;
    bool const first4 = g < E && h + 3 * NGC < j;
    if (h == 0 && g < E) pj_early = P[(size_t)j * ldp + g];
    if (first4) {
        x0 = vrow[(size_t)(h + 0 * NGC) * ldp]; x1 = vrow[(size_t)(h + 1 * NGC) * ldp];
        x2 = vrow[(size_t)(h + 2 * NGC) * ldp]; x3 = vrow[(size_t)(h + 3 * NGC) * ldp];
    }
    __syncthreads();
    double a = 0.0;
    if (g < E) {
        int l = h;
        if (first4) {
            a += x0 * s_w[l] + x1 * s_w[l + NGC] + x2 * s_w[l + 2 * NGC] + x3 * s_w[l + 3 * NGC];
            l += 4 * NGC;
        }
        for (; l + 3 * NGC < j; l += 4 * NGC)
            a += vrow[(size_t)(l + 0 * NGC) * ldp] * s_w[l] + vrow[(size_t)(l + 1 * NGC) * ldp] * s_w[l + NGC]
               + vrow[(size_t)(l + 2 * NGC) * ldp] * s_w[l + 2 * NGC] + vrow[(size_t)(l + 3 * NGC) * ldp] * s_w[l + 3 * NGC];
        for (; l < j; l += NGC) a += vrow[(size_t)l * ldp] * s_w[l];
    }
    if (h > 0) s_t[h - 1][r] = a;
    __syncthreads();
    if (h == 0) {
        double below = 0.0;
        if (g < E) {
            double pval = pj_early;
            if (j > 0) {                                                    // cpu.c:123-130
                #pragma unroll
                for (int q = 0; q < NGC - 1; q++) a += s_t[q][r];
                pval -= a;
                P[(size_t)j * ldp + g] = pval;
            }
            if (g > piv) below = pval;
        }
        s_p[r] = below;
        double ss = wave_sum(below * below);
        if (tid == 0) part_store(npart + blockIdx.x, ss);
    }
    __syncthreads();
    block_gemv_t_part<CTC>(V, ldp, g0, j, s_p, part + (size_t)blockIdx.x * MAXJ);
    slot_fold<CTC>(gridDim.x, j, part, acc + ACC_WVSUM, npart, acc + ACC_NRM, cnt);
}

// The big gemv: ypart[split][g] = sum_{c in split} A[g, c] * v[c],  g in [R0,E),
// c in [piv,E), v[piv] = 1, v[c] = scale * p''[c].
// One launch, 1-D grid:
//   blocks [0, nshadow)  : the "shadow" blocks (32 rows each) -- V(:,j) = v and
//                          VT(:,j) = tau (v - VT w_v); block 0 also publishes the scalars.
//   remaining blocks     : gemv tiles, 512 rows (each lane owns an aligned row pair,
//                          16-byte non-temporal loads: A is streamed once per column)
//                          x one column chunk; the 4 waves of a workgroup read 4 KiB
//                          contiguous per column.
// SHARD (block-column sharded reduction, hessenberg_sharded_device): the workgroup that finishes a row
// tile LAST adds the tile's column-split partials up into ysum -- the vector the ranks all-reduce --
// so that no launch of its own stands between the gemv and the collective.  Hand-off without
// fences (MI355X_MICROARCH.md, inter-workgroup visibility): every partial is stored and loaded
// with agent-scope (sc1) accesses, the stores are drained before the workgroup takes its ticket.
template <int UNROLL, bool ALIGNED, bool STREAM = true, bool SHARD = false>
__global__ __launch_bounds__(256)
void hess_gemv_kernel(double const *__restrict__ A, int ldA,
    double const *__restrict__ P, int R0, int E, int j, int cols_per_split, int ldp,
    int nshadow, int row_tiles,
    double *__restrict__ ypart, double *__restrict__ V, double *__restrict__ VT,
    double const *__restrict__ Y, double *__restrict__ t12,
    double const *__restrict__ acc, double *__restrict__ scal, int world, int rank,
    double *__restrict__ ysum = nullptr, int *__restrict__ tile_cnt = nullptr,
    HessExchange x = HessExchange{}, int seq = 0)
{
    __shared__ double s_wv[MAXJ], s_vrow[MAXJ], s_t[3][NGS][RBS + 1], s_scal[2];
    __shared__ int s_last;
    int const piv = R0 + j;
    double const *__restrict__ pcol = P + (size_t)j * ldp;

    if ((int)blockIdx.x < nshadow) {
        if (threadIdx.x == 0) {
            double scale, tau, beta;
            reflector_scalars(nrm_sum(acc), pcol[piv], scale, tau, beta);
            s_scal[0] = scale; s_scal[1] = tau;
            if (blockIdx.x == 0) {   // published for colA(j+1) / finish
                scal[4 * j + 0] = scale; scal[4 * j + 1] = tau; scal[4 * j + 2] = beta;
            }
        }
        __syncthreads();
        double const scale = s_scal[0];
        double const tau = s_scal[1];
        int const tid = threadIdx.x;
        int const r = tid & (RBS - 1), h = tid / RBS;
        int const g = R0 + blockIdx.x * RBS + r;
        for (int l = tid; l < j; l += 256) {
            double const vr = V[(size_t)l * ldp + piv];                 // V(piv, l)
            s_vrow[l] = vr;
            s_wv[l] = vr + scale * slot_sum(acc + ACC_WVSUM, l);
        }
        __syncthreads();
        // a = VT(g, 0:j) w_v (for the new column of VT); ya = Y(g, 0:j) w_v and pa = Y(g, 0:j) V(piv, 0:j)^T:
        // the two products colA(j+1) needs from the columns of Y that exist already (see colA)
        double a = 0.0, ya = 0.0, pa = 0.0;
        if (g < E) {
            double const *row = VT + g, *yrow = Y + g;
            int l = h;
            // (two columns per trip, four loads in flight: with more the kernel leaves the 64 VGPRs that let
            // eight waves of its STREAMING blocks share a SIMD -- measured: 80 VGPRs cost the gemv 4 %)
            for (; l + NGS < j; l += 2 * NGS) {
                double const x0 = row[(size_t)l * ldp], x1 = row[(size_t)(l + NGS) * ldp];
                double const y0 = yrow[(size_t)l * ldp], y1 = yrow[(size_t)(l + NGS) * ldp];
                a += x0 * s_wv[l] + x1 * s_wv[l + NGS];
                ya += y0 * s_wv[l] + y1 * s_wv[l + NGS];
                pa += y0 * s_vrow[l] + y1 * s_vrow[l + NGS];
            }
            for (; l < j; l += NGS) {
                double const y0 = yrow[(size_t)l * ldp];
                a += row[(size_t)l * ldp] * s_wv[l];
                ya += y0 * s_wv[l]; pa += y0 * s_vrow[l];
            }
        }
        s_t[0][h][r] = a; s_t[1][h][r] = ya; s_t[2][h][r] = pa;
        __syncthreads();
        if (h == 0 && g < E) {
            a = 0.0; ya = 0.0; pa = 0.0;
            #pragma unroll
            for (int q = 0; q < NGS; q++) { a += s_t[0][q][r]; ya += s_t[1][q][r]; pa += s_t[2][q][r]; }
            double v = g < piv ? 0.0 : (g == piv ? 1.0 : scale * pcol[g]);
            V[(size_t)j * ldp + g] = v;
            VT[(size_t)j * ldp + g] = tau * (v - a);
            t12[g] = ya; t12[ldp + g] = pa;
        }
        return;
    }

    int const b = blockIdx.x - nshadow;
    int const tile = b % row_tiles, split = b / row_tiles;
    int c_begin, c_end;
    if (rank < 0) {
        c_begin = piv + split * cols_per_split;
        c_end = min(E, c_begin + cols_per_split);
    } else {
        // block-column shards: this rank streams only the column blocks it owns
        // (block b = c / cols_per_split belongs to rank b % world); split s = its s-th block
        int const b0 = piv / cols_per_split;
        int const B = b0 + ((rank - b0 % world) + world) % world + split * world;
        c_begin = max(piv, B * cols_per_split);
        c_end = min(E, (B + 1) * cols_per_split);
    }
    double *yp = ypart + (size_t)split * ldp;
    if (ALIGNED) {
        // row tiles start on a 128-byte line of A (tiles that straddle lines stream slower); the
        // up to 15 rows above R0 are read and dropped
        int const g = (R0 & ~15) + tile * GEMV_ROWS + threadIdx.x * 2;
        if (!SHARD && g >= E) return;
        // y = A(:,piv) + scale * A(:,piv+1:) p''(piv+1:): the sum runs over the UNSCALED column (its
        // entries are uniform values that feed the FMAs straight from scalar registers: 60 VGPRs,
        // 8 waves per SIMD -- with the products scale*p'' held in vector registers it was 118 and the
        // 5 workgroups per CU of the launch did not fit at once), and the reflector scalars are
        // needed at the very end only, so no block waits for them before it starts to stream.
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0, f0 = 0.0, f1 = 0.0;
        if (g < E) {
        double const *a = A + (size_t)c_begin * ldA + g;
        int c = c_begin;
        if (c == piv) {
            d2 x = *reinterpret_cast<d2 const *>(a);
            f0 = x.x; f1 = x.y;
            c++; a += ldA;
        }
        for (; c + UNROLL <= c_end; c += UNROLL) {
            d2 x[UNROLL];
            #pragma unroll
            for (int u = 0; u < UNROLL; u++)
                x[u] = STREAM ? __builtin_nontemporal_load(reinterpret_cast<d2 const *>(a + (size_t)u * ldA))
                              : *reinterpret_cast<d2 const *>(a + (size_t)u * ldA);
            #pragma unroll
            for (int u = 0; u < UNROLL; u += 2) {
                double const v0 = pcol[c + u], v1 = pcol[c + u + 1];
                a0 += x[u].x * v0;     a1 += x[u].y * v0;
                b0 += x[u + 1].x * v1; b1 += x[u + 1].y * v1;
            }
            a += (size_t)UNROLL * ldA;
        }
        for (; c < c_end; c++) {
            d2 x = STREAM ? __builtin_nontemporal_load(reinterpret_cast<d2 const *>(a)) : *reinterpret_cast<d2 const *>(a);
            double const v0 = pcol[c];
            a0 += x.x * v0; a1 += x.y * v0;
            a += ldA;
        }
        }
        double scale, tau, beta;
        reflector_scalars(nrm_sum(acc), pcol[piv], scale, tau, beta);
        if (!SHARD) {
            if (g >= R0) yp[g] = f0 + scale * (a0 + b0);
            if (g + 1 >= R0 && g + 1 < E) yp[g + 1] = f1 + scale * (a1 + b1);
        } else {
            // Hand-off of the column-split partials to the workgroup that takes the tile's LAST ticket
            // (cdna_hip_programming.md section 6, Guideline 16, counter form): every partial is stored
            // write-through (sc1) and drained by its wave, the workgroup meets at its barrier, one lane
            // RELEASES at agent scope, takes the ticket, and the last arriver ACQUIRES at agent scope
            // before any wave of it loads a partial (sc1 loads as well).  (Rounds 3-4 relied on the sc1
            // accesses alone -- measured valid only for one workgroup per CU, this launch keeps ~5 resident;
            // that form is gone.)
            if (g >= R0 && g < E) __hip_atomic_store(yp + g, f0 + scale * (a0 + b0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g + 1 >= R0 && g + 1 < E) __hip_atomic_store(yp + g + 1, f1 + scale * (a1 + b1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int const nsplit = ((int)gridDim.x - nshadow) / row_tiles;
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the fence's own wait can be dropped by the compiler
                int const last = __hip_atomic_fetch_add(tile_cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsplit - 1;
                if (last) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // holds the barrier below until the invalidate is through
                }
                s_last = last;
            }
            __syncthreads();
            if (s_last) {
                #pragma unroll
                for (int q = 0; q < 2; q++) {
                    int const gg = g + q;
                    if (gg >= R0 && gg < E) {
                        double sum = 0.0;
                        for (int sp = 0; sp < nsplit; sp++)
                            sum += __hip_atomic_load(ypart + (size_t)sp * ldp + gg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (x.world == 0) ysum[gg] = sum;
                        else    // device-side exchange: this rank's slot on every rank (peer stores)
                            for (int r = 0; r < x.world; r++)
                                __hip_atomic_store(x.slots[r] + ((size_t)(seq & 1) * x.world + x.rank) * ldp + gg, sum,
                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
                if (threadIdx.x == 0) __hip_atomic_store(tile_cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (x.world) {
                    // the tile is out on every rank before its flag: drain, workgroup barrier, one release at
                    // system scope, then the flag stores (Guideline 16, flag form)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (threadIdx.x == 0) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        for (int r = 0; r < x.world; r++)
                            __hip_atomic_store(x.flags[r] + x.rank * HESS_MAX_ROW_TILES + tile, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
        }
    } else {
        // odd leading dimension / unaligned base: 8-byte loads
        for (int q = 0; q < 2; q++) {
            int const g = R0 + tile * GEMV_ROWS + q * 256 + threadIdx.x;
            if (g >= E) continue;
            double s = 0.0, f = 0.0;
            double const *a = A + (size_t)c_begin * ldA + g;
            int c = c_begin;
            if (c == piv) { f = *a; c++; a += ldA; }
            for (; c < c_end; c++, a += ldA) s += (*a) * pcol[c];
            double scale, tau, beta;
            reflector_scalars(nrm_sum(acc), pcol[piv], scale, tau, beta);
            yp[g] = f + scale * s;
        }
    }
}

__global__ void hess_copy_in_kernel(int R0, int E, int nb, int i,
    double const *__restrict__ A, int ldA, double *__restrict__ P, int ldp)
{
    int g = R0 + blockIdx.x * 256 + threadIdx.x;
    int j = blockIdx.y;
    if (g < E && j < nb) P[(size_t)j * ldp + g] = A[(size_t)(i + j) * ldA + g];
}
__global__ void hess_copy_out_kernel(int R0, int E, int nb, int i,
    double *__restrict__ A, int ldA, double const *__restrict__ P, int ldp)
{
    int g = R0 + blockIdx.x * 256 + threadIdx.x;
    int j = blockIdx.y;
    if (g < E && j < nb) A[(size_t)(i + j) * ldA + g] = P[(size_t)j * ldp + g];
}

// ---- workspace ----------------------------------------------------------------
struct HessWorkspace {
    int n = 0, nbmax = 0, ldp = 0, ysplits = 0;
    // YVW[b] = [ Y | V | W ] of one panel, three blocks of nb columns side by side with one leading
    // dimension: the fused trailing update A -= [Y V] [V' W]^T reads them as two ld x 2nb operands
    double *P = nullptr, *YVW[2] = {nullptr, nullptr}, *VT[2] = {nullptr, nullptr};
    double *S = nullptr, *W2 = nullptr;
    double *ypart = nullptr, *acc = nullptr, *scal = nullptr;
    double *part = nullptr, *npart = nullptr;   // ordered sums of the column kernels: a partial vector (and a scalar) per workgroup
    int *slot_cnt = nullptr;                    // their tickets: [0, NSLOT) colA, [NSLOT, 2 NSLOT) colC (self-resetting)
    double *t12 = nullptr;                      // [t1 | t2]: the products with the old columns of Y that colA(j+1) needs, formed in the shadow of gemv(j)
    static constexpr int MAX_ROW_TILES = HESS_MAX_ROW_TILES;    // row tiles of a gemv launch (512 rows each)
    int *tile_cnt = nullptr;                    // sharded gemv: arrivals per row tile (self-resetting)
    hipStream_t side = nullptr, main = nullptr;
    hipEvent_t entry = nullptr;
    hipEvent_t panel_done[2] = {nullptr, nullptr}, side_done[2] = {nullptr, nullptr};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> gemm_ev;        // per panel: (start, stop) of the critical and of the side updates
    std::vector<hipEvent_t> sample_ev;      // pairs (start, stop) around sampled gemv launches
    std::vector<double> sample_bytes;
    std::vector<hipEvent_t> comm_ev;        // sharded reduction: pairs around the timed collectives
    std::vector<int> comm_kind;
    std::vector<double> comm_payload;

    void release() {
        double **ptrs[] = {&P, &YVW[0], &YVW[1], &VT[0], &VT[1], &S, &W2, &ypart, &acc, &scal, &t12, &part, &npart};
        for (auto p : ptrs) if (*p) { SN_HIP_CHECK(hipFree(*p)); *p = nullptr; }
        if (tile_cnt) { SN_HIP_CHECK(hipFree(tile_cnt)); tile_cnt = nullptr; }
        if (slot_cnt) { SN_HIP_CHECK(hipFree(slot_cnt)); slot_cnt = nullptr; }
        n = nbmax = 0; ysplits = 0;
    }
    // streams and events go when the owning thread lets the workspace go (node finalize, the end of a team
    // thread): a stream holds a reference on a hardware queue -- or, created with a CU mask, the queue itself
    void destroy_streams() {
        auto kill = [](hipEvent_t &e) { if (e) { SN_HIP_CHECK(hipEventDestroy(e)); e = nullptr; } };
        if (main) { SN_HIP_CHECK(hipStreamDestroy(main)); main = nullptr; }
        if (side) { SN_HIP_CHECK(hipStreamDestroy(side)); side = nullptr; }
        kill(entry); kill(ev0); kill(ev1);
        for (int k = 0; k < 2; k++) { kill(panel_done[k]); kill(side_done[k]); }
        for (hipEvent_t &e : gemm_ev) kill(e);
        for (hipEvent_t &e : sample_ev) kill(e);
        for (hipEvent_t &e : comm_ev) kill(e);
        gemm_ev.clear(); sample_ev.clear(); comm_ev.clear();
    }
    void ensure(int n_, int nb_) {
        int const need_splits = std::max(MAX_SPLIT, divceil(n_, std::max(8, nb_)) + 1);
        if (n_ <= n && nb_ <= nbmax && need_splits <= ysplits) return;
        release();
        n = n_; nbmax = nb_;
        ldp = (int)roundup((size_t)n + GEMV_ROWS + 16, 128);
        size_t pan = (size_t)ldp * nbmax * sizeof(double);
        auto alloc = [](double **p, size_t bytes) {
            SN_HIP_CHECK(hipMalloc((void **)p, bytes));
            SN_HIP_CHECK(hipMemset(*p, 0, bytes));
        };
        alloc(&P, pan); alloc(&YVW[0], 3 * pan); alloc(&YVW[1], 3 * pan);
        alloc(&VT[0], pan); alloc(&VT[1], pan);
        alloc(&S, (size_t)nbmax * nbmax * sizeof(double)); alloc(&W2, pan);
        // one slice per column split of the gemv; the sharded path uses one per owned block column
        ysplits = need_splits;
        alloc(&ypart, (size_t)ysplits * ldp * sizeof(double));
        alloc(&acc, (size_t)ACC_TOTAL * sizeof(double));
        alloc(&scal, (size_t)4 * MAXJ * sizeof(double));
        alloc(&t12, (size_t)2 * ldp * sizeof(double));
        int const nwg_max = divceil(n, RB) + 1;
        alloc(&part, (size_t)nwg_max * MAXJ * sizeof(double));
        alloc(&npart, (size_t)nwg_max * sizeof(double));
        SN_HIP_CHECK(hipMalloc((void **)&slot_cnt, sizeof(int) * 2 * NSLOT));
        SN_HIP_CHECK(hipMemset(slot_cnt, 0, sizeof(int) * 2 * NSLOT));
        SN_HIP_CHECK(hipMalloc((void **)&tile_cnt, sizeof(int) * MAX_ROW_TILES));
        SN_HIP_CHECK(hipMemset(tile_cnt, 0, sizeof(int) * MAX_ROW_TILES));
        if (!side) {
            int lo = 0, hi = 0;
            SN_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));   // lo = least, hi = greatest
            // the latency-critical panel chain outranks the bulk GEMM updates
            make_stream(&main, true, hi);
            if (tuning().hess_side_cus > 0) {
                // experiment: the delayed updates confined to the first `hess_side_cus` CUs
                hipDeviceProp_t prop; int dev = 0;
                SN_HIP_CHECK(hipGetDevice(&dev)); SN_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
                int const ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
                std::vector<uint32_t> mask(words, 0u);
                for (int i = 0; i < std::min(ncu, tuning().hess_side_cus); i++) mask[i / 32] |= 1u << (i % 32);
                SN_HIP_CHECK(hipExtStreamCreateWithCUMask(&side, words, mask.data()));
            } else
            make_stream(&side, false, lo);
            SN_HIP_CHECK(hipEventCreateWithFlags(&entry, hipEventDisableTiming));
            for (int k = 0; k < 2; k++) {
                SN_HIP_CHECK(hipEventCreateWithFlags(&panel_done[k], hipEventDisableTiming));
                SN_HIP_CHECK(hipEventCreateWithFlags(&side_done[k], hipEventDisableTiming));
            }
            SN_HIP_CHECK(hipEventCreate(&ev0));
            SN_HIP_CHECK(hipEventCreate(&ev1));
        }
    }
};

// one workspace per host thread: the single-GPU path runs on the caller's thread, the in-process
// multi-GPU path (node_team.hip) on one persistent thread per device
static thread_local HessWorkspace g_ws;

void dgemm_release_workspace();
// (the scratch planes of the split-K products are keyed by stream: they go before the streams do)
void hessenberg_release_workspace() { dgemm_release_workspace(); g_ws.release(); g_ws.destroy_streams(); }

int hessenberg_panel_ld(int n, int) { return (int)roundup((size_t)n + GEMV_ROWS + 16, 128); }

static void choose_split(int m_rows, int ncols, int *nsplit, int *cps)
{
    int row_tiles = divceil(m_rows + 1, GEMV_ROWS);
    // ~4 streaming workgroups per CU (measured at n = 20000: 4.70 / 4.71 / 4.80 / 4.9 s for 512 /
    // 1024 / 1280 / 1792 workgroups); together with the shadow blocks of the launch they must all
    // be resident at once (8 workgroups of 256 threads per CU at 64 VGPRs)
    int const target_wgs = tuning().hess_wgs;
    int want = std::max(1, target_wgs / row_tiles);
    int const max_split = tuning().hess_max_split;
    int s = std::min({want, max_split, MAX_SPLIT, std::max(1, ncols / 16)});
    int c = divceil(ncols, s);
    c = (c + 15) / 16 * 16;
    s = divceil(ncols, c);
    *nsplit = std::max(1, s);
    *cps = c;
}

int hessenberg_device(hipStream_t caller, int n, int begin, int end, int panel_width,
    double *dA, int ldA, double *dQ, int ldQ, HessenbergTimings *tm)
{
    if (panel_width > MAXJ - 8) panel_width = MAXJ - 8;   // column kernels hold j < MAXJ in LDS
    HessWorkspace &ws = g_ws;
    ws.ensure(n, panel_width);
    int const ldp = ws.ldp;
    bool const aligned = (ldA % 2 == 0) && (((uintptr_t)dA) % 16 == 0);
    double gemv_bytes = 0.0, gemm_flops = 0.0, gemm_flops_main = 0.0, gemm_flops_fused = 0.0;
    long gemv_launches = 0;
    size_t nsampled = 0;
    int const sample_every = tm ? tm->sample_every : 0;
    bool const timed = tm != nullptr;
    ws.sample_bytes.clear();

    // everything runs on the library's own streams, fenced against the caller's stream
    hipStream_t s = ws.main;
    SN_HIP_CHECK(hipEventRecord(ws.entry, caller));
    SN_HIP_CHECK(hipStreamWaitEvent(s, ws.entry, 0));
    if (tm) SN_HIP_CHECK(hipEventRecord(ws.ev0, s));
    SN_HIP_CHECK(hipMemsetAsync(ws.slot_cnt, 0, sizeof(int) * 2 * NSLOT, s));    // (self-resetting; belt and braces)
    // profiling aid: stop after k panels (PMC runs cannot take 60 k dispatches); the result
    // is then a partial reduction and must not be used
    int const max_panels = tuning().hess_max_panels;
    int pcount = 0;
    for (int i = begin; i < end - 1 && pcount < max_panels; i += panel_width, pcount++) {
        int const nb = std::min(panel_width, end - i - 1);
        int const R0 = i + 1, E = end, m = E - R0;
        int const buf = pcount & 1;
        double *Y = ws.YVW[buf], *V = Y + (size_t)ldp * nb, *W = V + (size_t)ldp * nb, *VT = ws.VT[buf];
        int const nwg = divceil(m, RB);

        // the buffers of this slot were last used by the side stream two panels ago
        if (pcount >= 2) SN_HIP_CHECK(hipStreamWaitEvent(s, ws.side_done[buf], 0));

        SN_HIP_CHECK(hipMemsetAsync(ws.acc, 0, (size_t)ACC_TOTAL * sizeof(double), s));
        hipLaunchKernelGGL(hess_copy_in_kernel, dim3(divceil(m, 256), nb), dim3(256), 0, s,
            R0, E, nb, i, dA, ldA, ws.P, ldp);                                // core.c:451

        int nsplit = 1, cps = 0;
        for (int j = 0; j < nb; j++) {
            int const piv = R0 + j;
            if (j > 0)
                hipLaunchKernelGGL(hess_colA_kernel, dim3(nwg), dim3(CT), 0, s,
                    R0, E, j, ldp, ws.P, VT, Y, ws.ypart, nsplit, ws.t12, ws.acc, ws.scal, ws.part, ws.slot_cnt);
            int const ncols = E - piv;
            choose_split(m, ncols, &nsplit, &cps);
            long const cache_bytes = tuning().hess_cache_mb << 20;
            bool const streaming = aligned && (long)m * ncols * 8 > cache_bytes;
            hipLaunchKernelGGL(hess_colC_kernel, dim3(nwg), dim3(CTC), 0, s,
                R0, E, j, ldp, ws.P, V, ws.acc, ws.part, ws.npart, ws.slot_cnt + NSLOT);
            bool const sampled = sample_every > 0 && (gemv_launches % sample_every) == 0;
            if (sampled) {
                if (ws.sample_ev.size() < 2 * (nsampled + 1)) {
                    hipEvent_t a, b;
                    SN_HIP_CHECK(hipEventCreate(&a));
                    SN_HIP_CHECK(hipEventCreate(&b));
                    ws.sample_ev.push_back(a); ws.sample_ev.push_back(b);
                }
                SN_HIP_CHECK(hipEventRecord(ws.sample_ev[2 * nsampled], s));
            }
            int const row_tiles = divceil(E - (R0 & ~15), GEMV_ROWS);
            int const nshadow = divceil(m, RBS);
            dim3 grid(nshadow + row_tiles * nsplit);
            // once the trailing matrix fits the 256 MB Infinity Cache the next column re-reads part of
            // it from there: temporal loads (the streaming, non-temporal ones bypass the caches).
            // Measured: 4 % on the whole reduction at n = 6000, nothing at n = 20000.
            if (aligned && !streaming)
                hipLaunchKernelGGL((hess_gemv_kernel<16, true, false>), grid, dim3(256), 0, s,
                    dA, ldA, ws.P, R0, E, j, cps, ldp, nshadow, row_tiles, ws.ypart, V, VT, Y, ws.t12, ws.acc, ws.scal, 1, -1);
            else if (aligned)
                hipLaunchKernelGGL((hess_gemv_kernel<16, true>), grid, dim3(256), 0, s,
                    dA, ldA, ws.P, R0, E, j, cps, ldp, nshadow, row_tiles, ws.ypart, V, VT, Y, ws.t12, ws.acc, ws.scal, 1, -1);
            else
                hipLaunchKernelGGL((hess_gemv_kernel<16, false>), grid, dim3(256), 0, s,
                    dA, ldA, ws.P, R0, E, j, cps, ldp, nshadow, row_tiles, ws.ypart, V, VT, Y, ws.t12, ws.acc, ws.scal, 1, -1);
            if (sampled) {
                SN_HIP_CHECK(hipEventRecord(ws.sample_ev[2 * nsampled + 1], s));
                ws.sample_bytes.push_back(8.0 * (double)m * (double)ncols);
                nsampled++;
            }
            gemv_launches++;
            gemv_bytes += 8.0 * (double)m * (double)ncols;
        }
        hipLaunchKernelGGL(hess_finish_kernel, dim3(nwg), dim3(CT), 0, s,
            R0, E, nb, ldp, ws.P, V, Y, ws.ypart, nsplit, ws.acc, ws.scal);

        // ---- critical trailing updates (core.c:523-547), fused: with At the un-updated trailing
        // block, V' = V(i+nb:, :) and VT = V*T from the panel,
        //     W = (At - Y V'^T)^T VT = At^T VT - V' (Y^T VT)            (cpu.c:315-316, 373-384)
        //     At <- At - Y V'^T - V W^T = At - [Y V] [V' W]^T            (cpu.c:315, 433-435)
        // i.e. ONE read of At for W and ONE read-modify-write of At with k = 2 nb, instead of the
        // reference's right update, left product and left update (three passes, two of them RMW).
        if (timed) {
            while (ws.gemm_ev.size() < 6 * (size_t)(pcount + 1)) {
                hipEvent_t e; SN_HIP_CHECK(hipEventCreate(&e)); ws.gemm_ev.push_back(e);
            }
            SN_HIP_CHECK(hipEventRecord(ws.gemm_ev[6 * pcount + 0], s));
        }
        int const nt = E - (i + nb);
        if (nt > 0) {
            double *At = dA + (size_t)(i + nb) * ldA + R0;
            double *Wt = W + (i + nb), *Vp = V + (i + nb);       // rows of W, V' <-> columns of At
            dgemm(s, 'T', 'N', nt, nb, m, 1.0, At, ldA, VT + R0, ldp, 0.0, Wt, ldp);
            dgemm(s, 'T', 'N', nb, nb, m, 1.0, Y + R0, ldp, VT + R0, ldp, 0.0, ws.S, nb);
            dgemm(s, 'N', 'N', nt, nb, nb, -1.0, Vp, ldp, ws.S, nb, 1.0, Wt, ldp);
            if (timed) SN_HIP_CHECK(hipEventRecord(ws.gemm_ev[6 * pcount + 4], s));
            dgemm(s, 'N', 'T', m, nt, 2 * nb, -1.0, Y + R0, ldp, Vp, ldp, 1.0, At, ldA);
            if (timed) SN_HIP_CHECK(hipEventRecord(ws.gemm_ev[6 * pcount + 5], s));
            double const f = 6.0 * m * (double)nt * nb + 2.0 * nb * (double)nb * (m + nt);
            gemm_flops += f; gemm_flops_main += f; gemm_flops_fused += 4.0 * m * (double)nt * nb;
        } else if (timed) {
            SN_HIP_CHECK(hipEventRecord(ws.gemm_ev[6 * pcount + 4], s));
            SN_HIP_CHECK(hipEventRecord(ws.gemm_ev[6 * pcount + 5], s));
        }
        if (timed) SN_HIP_CHECK(hipEventRecord(ws.gemm_ev[6 * pcount + 1], s));
        // panel columns go back into A (core.c:317)
        hipLaunchKernelGGL(hess_copy_out_kernel, dim3(divceil(m, 256), nb), dim3(256), 0, s,
            R0, E, nb, i, dA, ldA, ws.P, ldp);
        SN_HIP_CHECK(hipEventRecord(ws.panel_done[buf], s));

        // ---- non-critical updates on the side stream (core.c:321-340) ----
        hipStream_t q = tuning().hess_noside ? s : ws.side;
        SN_HIP_CHECK(hipStreamWaitEvent(q, ws.panel_done[buf], 0));
        if (timed) SN_HIP_CHECK(hipEventRecord(ws.gemm_ev[6 * pcount + 2], q));
        {   // upper rows A(0:R0, R0:E) (I - V T V^T)
            double *X = dA + (size_t)R0 * ldA;
            dgemm(q, 'N', 'N', R0, nb, m, 1.0, X, ldA, VT + R0, ldp, 0.0, ws.W2, ldp);
            dgemm(q, 'N', 'T', R0, m, nb, -1.0, ws.W2, ldp, V + R0, ldp, 1.0, X, ldA);
            gemm_flops += 4.0 * R0 * (double)m * nb;
        }
        if (E < n) {   // columns right of end (core.c:330-336)
            int const nr = n - E;
            double *At2 = dA + (size_t)E * ldA + R0;
            dgemm(q, 'T', 'N', nr, nb, m, 1.0, At2, ldA, VT + R0, ldp, 0.0, ws.W2, ldp);
            dgemm(q, 'N', 'T', m, nr, nb, -1.0, V + R0, ldp, ws.W2, ldp, 1.0, At2, ldA);
            gemm_flops += 4.0 * nr * (double)m * nb;
        }
        if (dQ) {   // Q(:, R0:E) (I - V T V^T)
            double *X = dQ + (size_t)R0 * ldQ;
            dgemm(q, 'N', 'N', n, nb, m, 1.0, X, ldQ, VT + R0, ldp, 0.0, ws.W2, ldp);
            dgemm(q, 'N', 'T', n, m, nb, -1.0, ws.W2, ldp, V + R0, ldp, 1.0, X, ldQ);
            gemm_flops += 4.0 * n * (double)m * nb;
        }
        if (timed) SN_HIP_CHECK(hipEventRecord(ws.gemm_ev[6 * pcount + 3], q));
        SN_HIP_CHECK(hipEventRecord(ws.side_done[buf], q));
    }
    // join the side stream back into s
    for (int k = 0; k < 2 && k < pcount; k++)
        SN_HIP_CHECK(hipStreamWaitEvent(s, ws.side_done[k], 0));
    SN_HIP_CHECK(hipEventRecord(ws.ev1, s));
    SN_HIP_CHECK(hipStreamWaitEvent(caller, ws.ev1, 0));
    if (tm) {
        SN_HIP_CHECK(hipEventSynchronize(ws.ev1));
        float ms = 0.f;
        SN_HIP_CHECK(hipEventElapsedTime(&ms, ws.ev0, ws.ev1));
        tm->total_ms = ms;
        tm->gemv_bytes = gemv_bytes;
        tm->gemm_flops = gemm_flops;
        tm->gemm_flops_main = gemm_flops_main;
        tm->gemm_ms_main = tm->gemm_ms_side = tm->gemm_ms_fused = 0.0;
        tm->gemm_flops_fused = gemm_flops_fused;
        for (int p = 0; p < pcount; p++) {
            float t = 0.f;
            SN_HIP_CHECK(hipEventElapsedTime(&t, ws.gemm_ev[6 * p], ws.gemm_ev[6 * p + 1]));
            tm->gemm_ms_main += t;
            SN_HIP_CHECK(hipEventElapsedTime(&t, ws.gemm_ev[6 * p + 2], ws.gemm_ev[6 * p + 3]));
            tm->gemm_ms_side += t;
            SN_HIP_CHECK(hipEventElapsedTime(&t, ws.gemm_ev[6 * p + 4], ws.gemm_ev[6 * p + 5]));
            tm->gemm_ms_fused += t;
        }
        tm->gemv_launches = gemv_launches;
        tm->sampled_launches = (long)nsampled;
        tm->sampled_bytes = 0.0; tm->sampled_ms = 0.0;
        for (size_t k = 0; k < nsampled; k++) {
            float t = 0.f;
            SN_HIP_CHECK(hipEventElapsedTime(&t, ws.sample_ev[2 * k], ws.sample_ev[2 * k + 1]));
            tm->sampled_ms += t;
            tm->sampled_bytes += ws.sample_bytes[k];
        }
    }
    // a reduction cut short by the profiling switch is not a result: tell the caller
    return (max_panels < (1 << 30)) ? 1 : 0;
}

// ---- block-column sharded reduction over several GPUs (BASELINE config 4) -----
// One process per GPU; every rank holds full-size A and Q buffers but only MAINTAINS the
// column blocks of A it owns (block b of `cb` = panel_width columns belongs to rank
// b % world) and its contiguous row block of Q.  Per panel: the owner's panel columns are
// broadcast; the column chain (colA/colC and the shadow work) is replicated on every rank;
// the HBM-bound gemv is sharded by column block and the partial y vectors are summed by
// one all-reduce per column; the trailing updates touch owned blocks only; the update of
// the rows above the panel needs one all-reduce of W = A(0:i+1, .) V T per panel.  At the
// end the pieces are assembled on every rank (zero the unowned parts, all-reduce).
// Collectives are issued through callbacks (torch.distributed = RCCL in production) on
// buffers the caller allocated; everything runs on the caller's stream so that the
// collectives are ordered with the kernels.
__global__ void hess_ysum_kernel(int R0, int E, int nsplit, int ldp,
    double const *__restrict__ ypart, double *__restrict__ ysum)
{
    int g = R0 + blockIdx.x * 256 + threadIdx.x;
    if (g >= E) return;
    double s = 0.0;
    for (int k = 0; k < nsplit; k++) s += ypart[(size_t)k * ldp + g];
    ysum[g] = s;
}

// zero the columns of A this rank does not own (mode 0) / the rows of Q it does not own (mode 1)
__global__ void hess_zero_unowned_kernel(int mode, int n, double *__restrict__ X, int ld,
    int cb, int world, int rank, int row_lo, int row_hi)
{
    int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    for (int c = blockIdx.y; c < n; c += gridDim.y) {
        bool keep = mode == 0 ? ((c / cb) % world == rank) : (r >= row_lo && r < row_hi);
        if (!keep) X[(size_t)c * ld + r] = 0.0;
    }
}

// rows [r_lo, r_hi) of columns [c0, c0 + nc) of X <-> a contiguous (r_hi - r_lo) x nc block (dir 0: pack, 1: unpack)
__global__ void hess_pack_rows_kernel(int dir, int r_lo, int r_hi, int c0, int nc, double *__restrict__ X, int ld,
    double *__restrict__ buf)
{
    int const rows = r_hi - r_lo;
    int const r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    for (int c = blockIdx.y; c < nc; c += gridDim.y) {
        double *x = X + (size_t)(c0 + c) * ld + r_lo + r, *b = buf + (size_t)c * rows + r;
        if (dir == 0) *b = *x; else *x = *b;
    }
}

int hessenberg_sharded_device(hipStream_t s, int n, int panel_width,
    double *dA, int ldA, double *dQ, int ldQ,
    double *dYsum, double *dP, double *dW2, long w2_capacity,
    HessComm const &comm, HessenbergTimings *tm)
{
    if (panel_width > MAXJ - 8) panel_width = MAXJ - 8;
    int const world = comm.world, rank = comm.rank, cb = panel_width;
    HessWorkspace &ws = g_ws;
    if (ws.n != n) ws.release();        // the caller sized its buffers with hessenberg_panel_ld(n)
    ws.ensure(n, panel_width);
    int const ldp = ws.ldp;
    bool const aligned = (ldA % 2 == 0) && (((uintptr_t)dA) % 16 == 0);
    int const begin = 0, end = n;
    double gemv_bytes = 0.0, gemm_flops = 0.0;
    long gemv_launches = 0;
    // contiguous row block of Q owned by this rank (multiples of 128 rows)
    int const qchunk = (int)roundup(divceil(n, world), 128);
    int const q_lo = std::min(n, rank * qchunk), q_hi = std::min(n, (rank + 1) * qchunk);
    if ((long)n * panel_width > w2_capacity) return -1;
    // with one rank there is nothing to reduce: the column chain reads the gemv's partials directly
    bool const reduce_y = world > 1;
    // how the column-split partials of the sharded gemv are folded: 0 in the launch, by the last workgroup of
    // each row tile behind an agent-scope release / acquire; 2 by a launch of its own (the reproducer of
    // scratch/r5_oversub.sh)
    int const fold = tuning().hess_fold;
    // the per-column all-reduce of y on the device (one-process team, common.h HessExchange): column number
    // `seq` (counted across reductions: the flags are never reset) selects the parity of the slots
    bool const dev_x = comm.exchange != nullptr && reduce_y && aligned && fold != 2;
    HessExchange const x = dev_x ? *comm.exchange : HessExchange{};
    int seq = x.seq_base;
    if (reduce_y && aligned) SN_HIP_CHECK(hipMemsetAsync(ws.tile_cnt, 0, sizeof(int) * HessWorkspace::MAX_ROW_TILES, s));
    SN_HIP_CHECK(hipMemsetAsync(ws.slot_cnt, 0, sizeof(int) * 2 * NSLOT, s));
    // measurement (bench.py at N > 1): HIP events on the reduction's stream around every k-th gemv launch
    // and its all-reduce, around the per-panel collectives and the assembly
    int const sample_every = tm ? tm->sample_every : 0;
    size_t nsampled = 0, ncomm = 0;
    long allreduce_y_calls = 0;
    ws.sample_bytes.clear(); ws.comm_kind.clear(); ws.comm_payload.clear();
    auto sample_begin = [&]() {
        if (ws.sample_ev.size() < 2 * (nsampled + 1)) {
            hipEvent_t a, b;
            SN_HIP_CHECK(hipEventCreate(&a)); SN_HIP_CHECK(hipEventCreate(&b));
            ws.sample_ev.push_back(a); ws.sample_ev.push_back(b);
        }
        SN_HIP_CHECK(hipEventRecord(ws.sample_ev[2 * nsampled], s));
    };
    auto comm_begin = [&]() {
        if (ws.comm_ev.size() < 2 * (ncomm + 1)) {
            hipEvent_t a, b;
            SN_HIP_CHECK(hipEventCreate(&a)); SN_HIP_CHECK(hipEventCreate(&b));
            ws.comm_ev.push_back(a); ws.comm_ev.push_back(b);
        }
        SN_HIP_CHECK(hipEventRecord(ws.comm_ev[2 * ncomm], s));
    };
    auto comm_end = [&](int kind, double payload) {
        SN_HIP_CHECK(hipEventRecord(ws.comm_ev[2 * ncomm + 1], s));
        ws.comm_kind.push_back(kind); ws.comm_payload.push_back(payload); ncomm++;
    };
    bool const time_comm = sample_every > 0 && world > 1;

    if (tm) SN_HIP_CHECK(hipEventRecord(ws.ev0, s));
    int pcount = 0;
    for (int i = begin; i < end - 1; i += panel_width, pcount++) {
        int const nb = std::min(panel_width, end - i - 1);
        int const R0 = i + 1, E = end, m = E - R0;
        int const nwg = divceil(m, RB);
        int const owner = (i / cb) % world;
        int const buf = pcount & 1;
        // [Y | V | W] and VT of the panel alternate between two buffers: the Q update of panel p runs on
        // the side stream while panel p + 1 is factorised
        double *Ys = ws.YVW[buf], *V = Ys + (size_t)ldp * nb, *Ws = V + (size_t)ldp * nb, *VT = ws.VT[buf];
        if (pcount >= 2) SN_HIP_CHECK(hipStreamWaitEvent(s, ws.side_done[buf], 0));

        SN_HIP_CHECK(hipMemsetAsync(ws.acc, 0, (size_t)ACC_TOTAL * sizeof(double), s));
        // panel columns from their owner (the caller's dP is the panel buffer of every rank)
        hipLaunchKernelGGL(hess_copy_in_kernel, dim3(divceil(m, 256), nb), dim3(256), 0, s,
            R0, E, nb, i, dA, ldA, dP, ldp);
        if (world > 1) {
            if (time_comm) comm_begin();
            comm.broadcast(comm.ctx, 1, 0, (long)ldp * nb, owner);
            if (time_comm) comm_end(1, 8.0 * ldp * nb);
        }

        int nsplit = 0;
        for (int j = 0; j < nb; j++) {
            int const piv = R0 + j;
            double const *ysrc = reduce_y ? dYsum : ws.ypart;
            if (j > 0)
                hipLaunchKernelGGL(hess_colA_kernel, dim3(nwg), dim3(CT), 0, s,
                    R0, E, j, ldp, dP, VT, Ys, ysrc, reduce_y ? 1 : nsplit, ws.t12, ws.acc, ws.scal, ws.part, ws.slot_cnt, x, seq);
            hipLaunchKernelGGL(hess_colC_kernel, dim3(nwg), dim3(CTC), 0, s,
                R0, E, j, ldp, dP, V, ws.acc, ws.part, ws.npart, ws.slot_cnt + NSLOT);
            seq++;              // the number of THIS column's gemv
            // owned column blocks that intersect [piv, E): the splits of this rank's share of the gemv
            int const b0 = piv / cb;
            int const first = b0 + ((rank - b0 % world) + world) % world;
            int const last_block = (E - 1) / cb;
            nsplit = first > last_block ? 0 : (last_block - first) / world + 1;
            int const row_tiles = divceil(E - (R0 & ~15), GEMV_ROWS);
            int const nshadow = divceil(m, RBS);
            if (nsplit > ws.ysplits || row_tiles > HessWorkspace::MAX_ROW_TILES) return -2;
            dim3 grid(nshadow + row_tiles * nsplit);
            // this rank's share of the column's bytes: its owned blocks cut to [piv, E)
            double own_cols = 0.0;
            for (int B = first; B <= last_block; B += world)
                own_cols += std::min(E, (B + 1) * cb) - std::max(piv, B * cb);
            bool const sampled = sample_every > 0 && (gemv_launches % sample_every) == 0;
            if (sampled) sample_begin();
            if (aligned && reduce_y && fold != 2)
                hipLaunchKernelGGL((hess_gemv_kernel<16, true, true, true>), grid, dim3(256), 0, s,
                    dA, ldA, dP, R0, E, j, cb, ldp, nshadow, row_tiles, ws.ypart, V, VT, Ys, ws.t12, ws.acc, ws.scal,
                    world, rank, dYsum, ws.tile_cnt, x, seq);
            else if (aligned)
                hipLaunchKernelGGL((hess_gemv_kernel<16, true>), grid, dim3(256), 0, s,
                    dA, ldA, dP, R0, E, j, cb, ldp, nshadow, row_tiles, ws.ypart, V, VT, Ys, ws.t12, ws.acc, ws.scal,
                    world, rank);
            else
                hipLaunchKernelGGL((hess_gemv_kernel<16, false>), grid, dim3(256), 0, s,
                    dA, ldA, dP, R0, E, j, cb, ldp, nshadow, row_tiles, ws.ypart, V, VT, Ys, ws.t12, ws.acc, ws.scal,
                    world, rank);
            if (dev_x) {
                // nothing to launch: the gemv's last workgroups have written this rank's share to every rank.
                // (A rank without a column block right of the pivot publishes zeros.)
                if (nsplit == 0)
                    hipLaunchKernelGGL(hess_exchange_zero_kernel, dim3(row_tiles), dim3(256), 0, s, x, seq, R0, E, ldp);
                if (sampled) {
                    SN_HIP_CHECK(hipEventRecord(ws.sample_ev[2 * nsampled + 1], s));
                    ws.sample_bytes.push_back(8.0 * (double)m * own_cols); nsampled++;
                }
                allreduce_y_calls++;
            } else if (reduce_y) {
                if (nsplit == 0) SN_HIP_CHECK(hipMemsetAsync(dYsum + R0, 0, (size_t)m * sizeof(double), s));
                else if (!aligned || fold == 2)
                    hipLaunchKernelGGL(hess_ysum_kernel, dim3(divceil(m, 256)), dim3(256), 0, s,
                        R0, E, nsplit, ldp, ws.ypart, dYsum);
                if (sampled) {      // the fold is part of the launch (fold 0 / 1) or stands right behind it (fold 2)
                    SN_HIP_CHECK(hipEventRecord(ws.sample_ev[2 * nsampled + 1], s));
                    ws.sample_bytes.push_back(8.0 * (double)m * own_cols); nsampled++;
                }
                if (sampled && time_comm) comm_begin();
                comm.allreduce_sum(comm.ctx, 0, R0, (long)m);
                if (sampled && time_comm) comm_end(0, 8.0 * m);
                allreduce_y_calls++;
            } else if (sampled) {
                SN_HIP_CHECK(hipEventRecord(ws.sample_ev[2 * nsampled + 1], s));
                ws.sample_bytes.push_back(8.0 * (double)m * own_cols); nsampled++;
            }
            gemv_launches++;
            gemv_bytes += 8.0 * (double)m * own_cols;
        }
        hipLaunchKernelGGL(hess_finish_kernel, dim3(nwg), dim3(CT), 0, s,
            R0, E, nb, ldp, dP, V, Ys, reduce_y ? dYsum : ws.ypart, reduce_y ? 1 : nsplit, ws.acc, ws.scal, x, seq);

        // fused trailing update (hessenberg_device) on every run of adjacent owned blocks right of the panel
        // (one run when this rank owns everything; single blocks otherwise):
        //   W = At^T VT - V' (Y^T VT);   At <- At - [Y V] [V' W]^T              (core.c:523-547)
        bool have_s = false;
        for (int B = (i + nb) / cb; B * cb < E; ) {
            if (B % world != rank) { B++; continue; }
            int B1 = B + 1;
            while (B1 * cb < E && B1 % world == rank) B1++;
            int const c0 = std::max(B * cb, i + nb), c1 = std::min(E, B1 * cb), nt = c1 - c0;
            B = B1;
            if (nt <= 0) continue;
            if (!have_s) {
                dgemm(s, 'T', 'N', nb, nb, m, 1.0, Ys + R0, ldp, VT + R0, ldp, 0.0, ws.S, nb);
                gemm_flops += 2.0 * nb * (double)nb * m; have_s = true;
            }
            double *At = dA + (size_t)c0 * ldA + R0;
            double *Wt = Ws + c0, *Vp = V + c0;                  // rows of W, V' <-> columns of At
            dgemm(s, 'T', 'N', nt, nb, m, 1.0, At, ldA, VT + R0, ldp, 0.0, Wt, ldp);
            dgemm(s, 'N', 'N', nt, nb, nb, -1.0, Vp, ldp, ws.S, nb, 1.0, Wt, ldp);
            dgemm(s, 'N', 'T', m, nt, 2 * nb, -1.0, Ys + R0, ldp, Vp, ldp, 1.0, At, ldA);
            gemm_flops += 6.0 * m * (double)nt * nb + 2.0 * nb * (double)nb * nt;
        }
        // every rank stores the finished panel columns (rows >= R0 are final)
        hipLaunchKernelGGL(hess_copy_out_kernel, dim3(divceil(m, 256), nb), dim3(256), 0, s,
            R0, E, nb, i, dA, ldA, dP, ldp);

        SN_HIP_CHECK(hipEventRecord(ws.panel_done[buf], s));
        SN_HIP_CHECK(hipStreamWaitEvent(ws.side, ws.panel_done[buf], 0));
        // rows above the panel: W = sum over owned blocks A(0:R0, blk) VT(blk,:), all-reduce,
        // A(0:R0, blk) -= W V(blk,:)^T   (core.c:321-327; delayed there too).  With one rank: on the side
        // stream.  With several the all-reduce keeps it on the caller's stream: the collectives of one
        // communicator are issued on one stream, in one order.
        {
            hipStream_t const us = world == 1 ? ws.side : s;
            bool first_run = true;
            for (int B = R0 / cb; B * cb < E; ) {
                if (B % world != rank) { B++; continue; }
                int B1 = B + 1;
                while (B1 * cb < E && B1 % world == rank) B1++;
                int const c0 = std::max(B * cb, R0), c1 = std::min(E, B1 * cb), nt = c1 - c0;
                B = B1;
                if (nt <= 0) continue;
                dgemm(us, 'N', 'N', R0, nb, nt, 1.0, dA + (size_t)c0 * ldA, ldA, VT + c0, ldp, first_run ? 0.0 : 1.0, dW2, R0);
                first_run = false;
                gemm_flops += 2.0 * R0 * (double)nt * nb;
            }
            if (first_run) SN_HIP_CHECK(hipMemsetAsync(dW2, 0, (size_t)R0 * nb * sizeof(double), us));
            if (world > 1) {
                if (time_comm) comm_begin();
                comm.allreduce_sum(comm.ctx, 2, 0, (long)R0 * nb);
                if (time_comm) comm_end(2, 8.0 * R0 * nb);
            }
            for (int B = R0 / cb; B * cb < E; ) {
                if (B % world != rank) { B++; continue; }
                int B1 = B + 1;
                while (B1 * cb < E && B1 % world == rank) B1++;
                int const c0 = std::max(B * cb, R0), c1 = std::min(E, B1 * cb), nt = c1 - c0;
                B = B1;
                if (nt <= 0) continue;
                dgemm(us, 'N', 'T', R0, nt, nb, -1.0, dW2, R0, V + c0, ldp, 1.0, dA + (size_t)c0 * ldA, ldA);
                gemm_flops += 2.0 * R0 * (double)nt * nb;
            }
        }
        SN_HIP_CHECK(hipEventRecord(ws.panel_done[buf], s));
        // Q: this rank's row block (core.c:339-340), no communication: on the side stream, beside the next panel
        if (dQ && q_hi > q_lo) {
            int const rows = q_hi - q_lo;
            double *X = dQ + (size_t)R0 * ldQ + q_lo;
            dgemm(ws.side, 'N', 'N', rows, nb, m, 1.0, X, ldQ, VT + R0, ldp, 0.0, ws.W2, ldp);
            dgemm(ws.side, 'N', 'T', rows, m, nb, -1.0, ws.W2, ldp, V + R0, ldp, 1.0, X, ldQ);
            gemm_flops += 4.0 * rows * (double)m * nb;
        }
        SN_HIP_CHECK(hipEventRecord(ws.side_done[buf], ws.side));
    }
    for (int k = 0; k < 2 && k < pcount; k++)
        SN_HIP_CHECK(hipStreamWaitEvent(s, ws.side_done[k], 0));
    // assemble H and Q on every rank: every piece travels once, from its owner (an all-gather written as
    // broadcasts: the block columns of A are contiguous; the row blocks of Q go through dW2 in chunks of
    // columns) -- instead of zeroing the rest and all-reducing two full matrices
    if (world > 1) {
        double assembled = 0.0;
        if (time_comm) comm_begin();
        for (int B = 0; B * cb < n; B++) {
            int const c0 = B * cb, c1 = std::min(n, c0 + cb);
            comm.broadcast(comm.ctx, 3, (long)c0 * ldA, (long)(c1 - c0) * ldA, B % world);
            assembled += 8.0 * (c1 - c0) * ldA;
        }
        if (dQ) {
            for (int root = 0; root < world; root++) {
                int const r_lo = std::min(n, root * qchunk), r_hi = std::min(n, (root + 1) * qchunk), rows = r_hi - r_lo;
                if (rows <= 0) continue;
                int const chunk = (int)std::min<long>(n, w2_capacity / rows);
                for (int c0 = 0; c0 < n; c0 += chunk) {
                    int const nc = std::min(chunk, n - c0);
                    dim3 const grid(divceil(rows, 256), std::min(nc, 1024));
                    if (root == rank)
                        hipLaunchKernelGGL(hess_pack_rows_kernel, grid, dim3(256), 0, s, 0, r_lo, r_hi, c0, nc, dQ, ldQ, dW2);
                    comm.broadcast(comm.ctx, 2, 0, (long)rows * nc, root);
                    assembled += 8.0 * rows * nc;
                    if (root != rank)
                        hipLaunchKernelGGL(hess_pack_rows_kernel, grid, dim3(256), 0, s, 1, r_lo, r_hi, c0, nc, dQ, ldQ, dW2);
                }
            }
        }
        if (time_comm) comm_end(3, assembled);
    }
    if (tm) {
        SN_HIP_CHECK(hipEventRecord(ws.ev1, s));
        SN_HIP_CHECK(hipEventSynchronize(ws.ev1));
        float ms = 0.f;
        SN_HIP_CHECK(hipEventElapsedTime(&ms, ws.ev0, ws.ev1));
        tm->total_ms = ms;
        tm->gemv_bytes = gemv_bytes;
        tm->gemm_flops = gemm_flops;
        tm->gemv_launches = gemv_launches;
        tm->sampled_launches = (long)nsampled;
        tm->sampled_bytes = 0.0; tm->sampled_ms = 0.0;
        for (size_t k = 0; k < nsampled; k++) {
            float t = 0.f;
            SN_HIP_CHECK(hipEventElapsedTime(&t, ws.sample_ev[2 * k], ws.sample_ev[2 * k + 1]));
            tm->sampled_ms += t; tm->sampled_bytes += ws.sample_bytes[k];
        }
        for (int k = 0; k < 4; k++) { tm->comm_ms[k] = 0.0; tm->comm_bytes[k] = 0.0; tm->comm_calls[k] = 0; }
        for (size_t k = 0; k < ncomm; k++) {
            float t = 0.f;
            SN_HIP_CHECK(hipEventElapsedTime(&t, ws.comm_ev[2 * k], ws.comm_ev[2 * k + 1]));
            int const kind = ws.comm_kind[k];
            tm->comm_ms[kind] += t; tm->comm_bytes[kind] += ws.comm_payload[k]; tm->comm_calls[kind]++;
        }
        tm->allreduce_y_calls = allreduce_y_calls;
    }
    return 0;
}

} // namespace sn

#ifdef SN_TEST_HOOKS   // compiled into libstarneig_amd_test.so only (csrc/Makefile)
// scratch/hess_panel_probe.py: the panel factors [V | VT] and the reflector scalars of the panel
// that was factorised last (buffer slot `buf`), copied to the host as ldp x nb column-major blocks
extern "C" __attribute__((visibility("default")))
int sn_internal_hess_panel_factors(int buf, int nb, double *V, double *VT, double *scal, int *ldp_out)
{
    using namespace sn;
    HessWorkspace &ws = g_ws;
    if (!ws.P || nb > ws.nbmax) return -1;
    SN_HIP_CHECK(hipDeviceSynchronize());
    size_t const pan = (size_t)ws.ldp * nb;
    double *Y = ws.YVW[buf];
    SN_HIP_CHECK(hipMemcpy(V, Y + pan, pan * 8, hipMemcpyDeviceToHost));
    SN_HIP_CHECK(hipMemcpy(VT, ws.VT[buf], pan * 8, hipMemcpyDeviceToHost));
    SN_HIP_CHECK(hipMemcpy(scal, ws.scal, (size_t)4 * nb * 8, hipMemcpyDeviceToHost));
    *ldp_out = ws.ldp;
    return 0;
}
#endif  // SN_TEST_HOOKS

