// The window kernels of the host side (schur_host.hip) as a serial chain plus a team of helpers.
//
// Every step of the sequential window algorithms (double-shift QR sweep, swap of adjacent diagonal
// blocks, re-reduction to Hessenberg form: LAPACK dlahqr / dlaexc / dgehd2, which the reference calls
// from schur/cpu_utils.c:2248-2309, :2837-3046) applies one small orthogonal factor G to a few rows of
// T from the left, to the same few columns of T from the right, and to those columns of Z.  Only a
// part of that is read again by the NEXT step: the rows inside the block still being worked on and
// the columns below the rows already finished.  The rest --
//      Z(:, k:k+nr)            (accumulation; nothing reads Z before the end)
//      T(k:k+nr, from:n)       (columns right of the active block)
//      T(0:upto, k:k+nr)       (rows above it)
// -- is written to an operation log and applied, in log order, by helper threads that each own a
// fixed slice of the data (Z: a contiguous range of rows; T: every nt-th column right of `from`, every
// nt-th chunk of eight rows above `upto`).  Every element therefore sees the same factors in the same
// order, with the same arithmetic, as in the serial loop: the results do not depend on timing, and they
// are bit-identical to the serial path (which runs the SAME kernels over the whole ranges).
// The log is published once per sweep / swap, not per factor: one shared cache line moves per batch.
#pragma once
#include <atomic>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <vector>
#include <cstring>
#include <algorithm>
#include <immintrin.h>
#include <sched.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/file.h>
#include <cstdio>

namespace sn { namespace host {

struct Op {
    int k;              // first row / column G acts on
    short nr;           // how many (2..4 for a short reflector, any length for a long one, 2 for a rotation)
    short kind;         // 0: reflector I - d [1 a b c]^T [1 a b c];  1: rotation (a = cs, b = sn);  2: long reflector, v behind `a`
    int from;           // helpers: T(k:k+nr, from:n) from the left
    int upto;           // helpers: T(0:upto, k:k+nr) from the right
    double a, b, c, d;
};
static_assert(sizeof(Op) == 48, "three operations in nine quarter lines");

static inline double const *op_vector(Op const &op) { double const *p; std::memcpy(&p, &op.a, sizeof p); return p; }
static inline void op_set_vector(Op &op, double const *p) { std::memcpy(&op.a, &p, sizeof p); }

// ---- the kernels ---------------------------------------------------------------------------------
// Row operations on a column-major matrix touch nr adjacent entries of every column: the loops cannot
// vectorise, and without help the compiler serialises them (a store to column j might alias the load
// of column j+1 for all it knows).  Four columns are loaded before any is stored.
// X(k:k+nr, j) <- G^T X(k:k+nr, j) for j = j0, j0+step, ... < j1
static inline void op_rows(double *X, int ldx, Op const &op, int j0, int j1, int step)
{
    if (j0 >= j1) return;
    size_t const cs = (size_t)step * ldx;
    double *p = X + (size_t)j0 * ldx + op.k;
    int cnt = (j1 - j0 + step - 1) / step;
    if (op.kind == 1) {
        double const c = op.a, s = op.b;
        for (; cnt >= 4; cnt -= 4, p += 4 * cs) {
            double *p1 = p + cs, *p2 = p1 + cs, *p3 = p2 + cs;
            double x0 = p[0], y0 = p[1], x1 = p1[0], y1 = p1[1], x2 = p2[0], y2 = p2[1], x3 = p3[0], y3 = p3[1];
            p[0] = c * x0 + s * y0; p[1] = c * y0 - s * x0;
            p1[0] = c * x1 + s * y1; p1[1] = c * y1 - s * x1;
            p2[0] = c * x2 + s * y2; p2[1] = c * y2 - s * x2;
            p3[0] = c * x3 + s * y3; p3[1] = c * y3 - s * x3;
        }
        for (; cnt > 0; cnt--, p += cs) { double x = p[0], y = p[1]; p[0] = c * x + s * y; p[1] = c * y - s * x; }
        return;
    }
    if (op.kind == 2) {
        double const *v = op_vector(op); double const tau = op.d; int const len = op.nr;
        for (; cnt > 0; cnt--, p += cs) {
            double *__restrict__ x = p;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0, s6 = 0.0, s7 = 0.0;   // fixed partial sums: vectorises
            int i = 0;
            for (; i + 8 <= len; i += 8) {
                s0 += v[i] * x[i]; s1 += v[i + 1] * x[i + 1]; s2 += v[i + 2] * x[i + 2]; s3 += v[i + 3] * x[i + 3];
                s4 += v[i + 4] * x[i + 4]; s5 += v[i + 5] * x[i + 5]; s6 += v[i + 6] * x[i + 6]; s7 += v[i + 7] * x[i + 7];
            }
            for (; i < len; i++) s0 += v[i] * x[i];
            double const s = (((s0 + s4) + (s1 + s5)) + ((s2 + s6) + (s3 + s7))) * tau;
            for (i = 0; i < len; i++) x[i] -= s * v[i];
        }
        return;
    }
    double const v2 = op.a, v3 = op.b, v4 = op.c, t1 = op.d, t2 = t1 * v2, t3 = t1 * v3, t4 = t1 * v4;
    if (op.nr == 3) {
        for (; cnt >= 4; cnt -= 4, p += 4 * cs) {
            double *p1 = p + cs, *p2 = p1 + cs, *p3 = p2 + cs;
            double a0 = p[0], a1 = p[1], a2 = p[2], b0 = p1[0], b1 = p1[1], b2 = p1[2];
            double c0 = p2[0], c1 = p2[1], c2 = p2[2], d0 = p3[0], d1 = p3[1], d2 = p3[2];
            double sa = a0 + v2 * a1 + v3 * a2, sb = b0 + v2 * b1 + v3 * b2;
            double sc = c0 + v2 * c1 + v3 * c2, sd = d0 + v2 * d1 + v3 * d2;
            p[0] = a0 - sa * t1; p[1] = a1 - sa * t2; p[2] = a2 - sa * t3;
            p1[0] = b0 - sb * t1; p1[1] = b1 - sb * t2; p1[2] = b2 - sb * t3;
            p2[0] = c0 - sc * t1; p2[1] = c1 - sc * t2; p2[2] = c2 - sc * t3;
            p3[0] = d0 - sd * t1; p3[1] = d1 - sd * t2; p3[2] = d2 - sd * t3;
        }
        for (; cnt > 0; cnt--, p += cs) {
            double a0 = p[0], a1 = p[1], a2 = p[2], sa = a0 + v2 * a1 + v3 * a2;
            p[0] = a0 - sa * t1; p[1] = a1 - sa * t2; p[2] = a2 - sa * t3;
        }
    } else if (op.nr == 2) {
        for (; cnt >= 4; cnt -= 4, p += 4 * cs) {
            double *p1 = p + cs, *p2 = p1 + cs, *p3 = p2 + cs;
            double a0 = p[0], a1 = p[1], b0 = p1[0], b1 = p1[1], c0 = p2[0], c1 = p2[1], d0 = p3[0], d1 = p3[1];
            double sa = a0 + v2 * a1, sb = b0 + v2 * b1, sc = c0 + v2 * c1, sd = d0 + v2 * d1;
            p[0] = a0 - sa * t1; p[1] = a1 - sa * t2; p1[0] = b0 - sb * t1; p1[1] = b1 - sb * t2;
            p2[0] = c0 - sc * t1; p2[1] = c1 - sc * t2; p3[0] = d0 - sd * t1; p3[1] = d1 - sd * t2;
        }
        for (; cnt > 0; cnt--, p += cs) {
            double a0 = p[0], a1 = p[1], sa = a0 + v2 * a1;
            p[0] = a0 - sa * t1; p[1] = a1 - sa * t2;
        }
    } else {
        for (; cnt >= 2; cnt -= 2, p += 2 * cs) {
            double *p1 = p + cs;
            double a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], b0 = p1[0], b1 = p1[1], b2 = p1[2], b3 = p1[3];
            double sa = a0 + v2 * a1 + v3 * a2 + v4 * a3, sb = b0 + v2 * b1 + v3 * b2 + v4 * b3;
            p[0] = a0 - sa * t1; p[1] = a1 - sa * t2; p[2] = a2 - sa * t3; p[3] = a3 - sa * t4;
            p1[0] = b0 - sb * t1; p1[1] = b1 - sb * t2; p1[2] = b2 - sb * t3; p1[3] = b3 - sb * t4;
        }
        for (; cnt > 0; cnt--, p += cs) {
            double a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], sa = a0 + v2 * a1 + v3 * a2 + v4 * a3;
            p[0] = a0 - sa * t1; p[1] = a1 - sa * t2; p[2] = a2 - sa * t3; p[3] = a3 - sa * t4;
        }
    }
}

// X(r0:r1, k:k+nr) <- X(r0:r1, k:k+nr) G: distinct columns, contiguous rows: the loops vectorise.
// `w` (r1 doubles) is scratch for the long reflectors only.
// (Compiled twice: the library is built for x86-64-v3, the 512-bit clone is picked at load time where
// the processor has it -- these loops are a quarter of the serial chain's time.)
#if !defined(__HIP_DEVICE_COMPILE__) && !defined(SN_NO_TARGET_CLONES)    // (ifunc resolvers run before a sanitizer is up)
__attribute__((target_clones("arch=x86-64-v4", "default")))
#endif
static void op_cols(double *X, int ldx, Op const &op, int r0, int r1, double *w = nullptr)
{
    if (r0 >= r1) return;
    double *__restrict__ c0 = X + (size_t)op.k * ldx, *__restrict__ c1 = c0 + ldx;
    if (op.kind == 1) {
        double const c = op.a, s = op.b;
        for (int i = r0; i < r1; i++) { double x = c0[i], y = c1[i]; c0[i] = c * x + s * y; c1[i] = c * y - s * x; }
        return;
    }
    if (op.kind == 2) {
        double const *v = op_vector(op); double const tau = op.d; int const len = op.nr;
        double *__restrict__ ww = w;
        for (int i = r0; i < r1; i++) ww[i] = 0.0;
        for (int j = 0; j < len; j++) {
            double const *__restrict__ x = c0 + (size_t)j * ldx; double const vj = v[j];
            for (int i = r0; i < r1; i++) ww[i] += x[i] * vj;
        }
        for (int j = 0; j < len; j++) {
            double *__restrict__ x = c0 + (size_t)j * ldx; double const tv = tau * v[j];
            for (int i = r0; i < r1; i++) x[i] -= ww[i] * tv;
        }
        return;
    }
    double const v2 = op.a, v3 = op.b, v4 = op.c, t1 = op.d, t2 = t1 * v2, t3 = t1 * v3, t4 = t1 * v4;
    if (op.nr == 3) {
        double *__restrict__ c2 = c1 + ldx;
        for (int i = r0; i < r1; i++) {
            double sum = c0[i] + v2 * c1[i] + v3 * c2[i];
            c0[i] -= sum * t1; c1[i] -= sum * t2; c2[i] -= sum * t3;
        }
    } else if (op.nr == 2) {
        for (int i = r0; i < r1; i++) {
            double sum = c0[i] + v2 * c1[i];
            c0[i] -= sum * t1; c1[i] -= sum * t2;
        }
    } else {
        double *__restrict__ c2 = c1 + ldx, *__restrict__ c3 = c2 + ldx;
        for (int i = r0; i < r1; i++) {
            double sum = c0[i] + v2 * c1[i] + v3 * c2[i] + v4 * c3[i];
            c0[i] -= sum * t1; c1[i] -= sum * t2; c2[i] -= sum * t3; c3[i] -= sum * t4;
        }
    }
}

// ---- where the threads of a team sit --------------------------------------------------------------
// The helpers read what the calling thread has just written, and spin while they wait: the team wants
// one core per thread, all under one L3, and nobody else's spinning thread on them.  A caller that
// drifts onto a helper's core runs at a fraction of its speed, and so do two processes of one node
// (the ranks of a sharded reduction) that pick the same cores.  So cores are CLAIMED: an exclusive
// flock on /dev/shm/starneig_amd.core<N>, held for the session and dropped by the kernel if the process
// dies.  The first L3 domain (the caller's own first) with enough free cores in the allowed set wins;
// the caller moves to the first claimed core (its own if that was free).  Best effort throughout:
// without /sys or /dev/shm, or with too few free cores, nobody is pinned and nothing is claimed.
struct Seats {
    struct Topology {
        std::vector<int> core_of;                   // cpu -> lowest cpu of its core (-1: unknown)
        std::vector<std::vector<int>> domain;       // L3 domain -> its cores (each by its lowest cpu)
        std::vector<int> domain_of;                 // cpu -> L3 domain
        Topology()
        {
            auto cpu_list = [](int cpu, char const *what, std::vector<int> &out) {
                char path[160];
                snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/%s", cpu, what);
                FILE *f = fopen(path, "r");
                if (!f) return false;
                for (;;) {
                    int a, b;
                    if (fscanf(f, "%d", &a) != 1) break;
                    b = a;
                    int ch = fgetc(f);
                    if (ch == '-') { if (fscanf(f, "%d", &b) != 1) break; ch = fgetc(f); }
                    for (int c = a; c <= b; c++) out.push_back(c);
                    if (ch != ',') break;
                }
                fclose(f);
                return !out.empty();
            };
            int const ncpu = std::min((int)std::thread::hardware_concurrency(), (int)CPU_SETSIZE);
            core_of.assign(ncpu, -1); domain_of.assign(ncpu, -1);
            for (int c = 0; c < ncpu; c++) {
                if (domain_of[c] >= 0) continue;
                std::vector<int> l3;
                if (!cpu_list(c, "cache/index3/shared_cpu_list", l3)) continue;
                int const d = (int)domain.size();
                domain.emplace_back();
                for (int x : l3) {
                    if (x < 0 || x >= ncpu) continue;
                    domain_of[x] = d;
                    if (core_of[x] < 0) {
                        std::vector<int> sib;
                        if (!cpu_list(x, "topology/thread_siblings_list", sib)) sib.assign(1, x);
                        int const lead = *std::min_element(sib.begin(), sib.end());
                        for (int y : sib) if (y >= 0 && y < ncpu) core_of[y] = lead;
                    }
                    if (core_of[x] == x) domain[d].push_back(x);
                }
            }
        }
    };
    std::vector<int> held;      // file descriptors of the claims

    static int claim_core(int core)
    {
        char path[64];
        snprintf(path, sizeof path, "/dev/shm/starneig_amd.core%d", core);
        int const fd = ::open(path, O_CREAT | O_RDWR | O_CLOEXEC, 0666);
        if (fd < 0) return -1;
        if (flock(fd, LOCK_EX | LOCK_NB) != 0) { ::close(fd); return -1; }
        return fd;
    }
    void release() { for (int fd : held) ::close(fd); held.clear(); }
    // `want` cores under one L3, seat[0] for the caller
    bool claim(int want, cpu_set_t const &allowed, int *seat)
    {
        static Topology const topo;
        int const self = sched_getcpu();
        if (topo.domain.empty() || self < 0 || self >= (int)topo.core_of.size() || topo.domain_of[self] < 0) return false;
        int const home = topo.domain_of[self], nd = (int)topo.domain.size();
        for (int t = 0; t < nd; t++) {
            int const d = (home + t) % nd;
            std::vector<int> order;
            if (d == home) order.push_back(topo.core_of[self]);
            for (int c : topo.domain[d]) if (!(d == home && c == topo.core_of[self])) order.push_back(c);
            int got = 0;
            for (int c : order) {
                if (!CPU_ISSET(c, &allowed)) continue;
                int const fd = claim_core(c);
                if (fd < 0) continue;
                held.push_back(fd);
                seat[got++] = (c == topo.core_of[self]) ? self : c;
                if (got == want) return true;
            }
            release();
        }
        return false;
    }
    ~Seats() { release(); }
};

// ---- the team ------------------------------------------------------------------------------------
struct Team {
    static constexpr unsigned CAP = 1u << 13;
    static constexpr int MAXH = 8;
    static constexpr int CHUNK = 8;                     // rows of T above `upto` are dealt out in chunks of this many

    Op ops[CAP];
    alignas(64) std::atomic<unsigned> head{0};          // published operations
    struct alignas(64) Tail { std::atomic<unsigned> v{0}; } tail[MAXH];
    alignas(64) unsigned pending = 0;                   // written, not yet published (calling thread only)
    unsigned floor_seen = 0;                            // lowest tail at the last look
    // the matrices of the current call: set while every helper is idle, read after an acquire of `head`
    double *T = nullptr, *Z = nullptr;
    int ldt = 0, ldz = 0, n = 0;
    int nz = 0, nt = 0;                                 // helpers [0, nz) own rows of Z, [nz, nz + nt) own slices of T
    std::thread th[MAXH];
    std::mutex mu;
    std::condition_variable cv;
    // helpers that found no work for ~IDLE_SPINS pauses (the caller is blocked on the GPU: a sweep takes
    // tens of milliseconds) stop spinning and wait here until the next publish()
    static constexpr unsigned IDLE_SPINS = 1u << 16;
    alignas(64) std::atomic<int> sleepers{0};
    std::condition_variable cv_work;
    bool session = false, quit = false, pinned = false;
    cpu_set_t caller_mask;
    Seats seats;
    int started = 0, parked = 0;        // (parked: under `mu`)

    void begin(double *T_, int ldt_, double *Z_, int ldz_, int n_)
    {
        wait_all();
        T = T_; ldt = ldt_; Z = Z_; ldz = ldz_; n = n_;
    }
    inline void log(Op const &op)
    {
        if (pending - floor_seen >= CAP) {              // the ring is full as far as this thread knows
            publish();
            for (;;) {
                unsigned lo = pending;
                for (int h = 0; h < nz + nt; h++) { unsigned t = tail[h].v.load(std::memory_order_acquire); if ((int)(t - lo) < 0) lo = t; }
                floor_seen = lo;
                if (pending - lo < CAP) break;
                _mm_pause();
            }
        }
        ops[pending % CAP] = op;
        pending++;
    }
    inline void publish()
    {
        // (seq_cst on both sides: the store of head and the load of sleepers must not pass each other,
        // nor the helper's increment of sleepers and its last look at head)
        head.store(pending, std::memory_order_seq_cst);
        if (sleepers.load(std::memory_order_seq_cst) != 0) { std::lock_guard<std::mutex> lk(mu); cv_work.notify_all(); }
    }
    void wait(int h0, int h1)
    {
        publish();
        for (int h = h0; h < h1; h++)
            while (tail[h].v.load(std::memory_order_acquire) != pending) _mm_pause();
    }
    void wait_t() { wait(nz, nz + nt); }                // T is whole again (Z may still be behind)
    // the helpers of T have passed log position `idx` (which must have been published)
    inline void wait_t_index(unsigned idx)
    {
        for (int h = nz; h < nz + nt; h++)
            while ((int)(tail[h].v.load(std::memory_order_acquire) - idx) < 0) _mm_pause();
    }
    void wait_all() { wait(0, nz + nt); }

    void run(int me)
    {
        std::vector<double> w;
        for (;;) {
            {   // parked between sessions
                std::unique_lock<std::mutex> lk(mu);
                parked++;
                cv.wait(lk, [&] { return (session && me < nz + nt) || quit; });
                parked--;
                if (quit) return;
            }
            unsigned idle = 0;
            unsigned t = tail[me].v.load(std::memory_order_relaxed);
            for (;;) {
                unsigned const h = head.load(std::memory_order_acquire);
                if (t != h) {
                    double *const Tm = T, *const Zm = Z; int const lt = ldt, lz = ldz, nn = n;
                    if ((int)w.size() < nn) w.resize(nn);
                    if (me < nz) {
                        // rows [r0, r1) of Z, multiples of 8 apart
                        int const blocks = (nn + 7) / 8;
                        int const r0 = std::min(nn, 8 * (int)((long)blocks * me / nz)), r1 = std::min(nn, 8 * (int)((long)blocks * (me + 1) / nz));
                        for (; t != h; t++) op_cols(Zm, lz, ops[t % CAP], r0, r1, w.data());
                    } else {
                        int const q = me - nz;
                        for (; t != h; t++) {
                            Op const &op = ops[t % CAP];
                            if (op.from < nn) {
                                int j0 = op.from + ((q - op.from) % nt + nt) % nt;
                                op_rows(Tm, lt, op, j0, nn, nt);
                            }
                            for (int c = q * CHUNK; c < op.upto; c += nt * CHUNK)
                                op_cols(Tm, lt, op, c, std::min(c + CHUNK, op.upto), w.data());
                        }
                    }
                    tail[me].v.store(t, std::memory_order_release);
                    idle = 0;
                } else {
                    _mm_pause();
                    ++idle;
                    if ((idle & 8191) == 0) {
                        std::unique_lock<std::mutex> lk(mu);
                        if (!session) break;
                        if (idle >= IDLE_SPINS) {
                            sleepers.fetch_add(1, std::memory_order_seq_cst);
                            cv_work.wait(lk, [&] { return head.load(std::memory_order_seq_cst) != t || !session; });
                            sleepers.fetch_sub(1, std::memory_order_seq_cst);
                            idle = 0;
                            if (!session) break;
                        }
                    }
                }
            }
        }
    }
    // helpers: `count` threads, a third of them (at least one) on Z, the others on T
    void open(int count)
    {
        count = std::max(2, std::min(count, (int)MAXH));
        if (started && count != nz + nt) {
            // another division of the work: every helper of the last session has to be parked first (one
            // that is still spinning would go on with its old share next to the new owners of it)
            for (;;) {
                { std::lock_guard<std::mutex> lk(mu); if (parked == started) break; }
                std::this_thread::yield();
            }
        }
        std::lock_guard<std::mutex> lk(mu);
        nz = std::max(1, count / 3); nt = count - nz;        // the far columns of the active block are the larger share
        int seat[MAXH + 1];
        cpu_set_t allowed;
        pinned = pthread_getaffinity_np(pthread_self(), sizeof allowed, &allowed) == 0 && seats.claim(count + 1, allowed, seat);
        for (int h = 0; h < MAXH; h++) tail[h].v.store(pending, std::memory_order_relaxed);   // every helper starts level with the log
        for (; started < count; started++) th[started] = std::thread([this, me = started] { run(me); });
        if (pinned) {
            caller_mask = allowed;
            auto pin = [](pthread_t t, int cpu) { cpu_set_t one; CPU_ZERO(&one); CPU_SET(cpu, &one); pthread_setaffinity_np(t, sizeof one, &one); };
            pin(pthread_self(), seat[0]);
            for (int h = 0; h < count; h++) pin(th[h].native_handle(), seat[h + 1]);
        }
        session = true;
        cv.notify_all();
    }
    void close()
    {
        wait_all();
        if (pinned) { pthread_setaffinity_np(pthread_self(), sizeof caller_mask, &caller_mask); seats.release(); pinned = false; }
        std::lock_guard<std::mutex> lk(mu);
        session = false;
        cv_work.notify_all();
    }
    ~Team()
    {
        if (!started) return;
        { std::lock_guard<std::mutex> lk(mu); quit = true; session = false; }
        cv.notify_all();
        cv_work.notify_all();
        for (int h = 0; h < started; h++) if (th[h].joinable()) th[h].join();
    }
};

// What a window kernel sees: "apply G now where the chain needs it, and see to the rest".
struct Applier {
    double *T; int ldt; double *Z; int ldz; int n;
    Team *team = nullptr;       // null: everything is applied at once, on the calling thread
    double *z0 = nullptr;       // if set: a private copy of Z(0, :) kept current on the calling thread
    std::vector<double> w;      // scratch of the long reflectors

    // The calling thread's share: T(k:k+nr, c0:from) from the left, T(upto:r1, k:k+nr) from the right.
    inline void emit(Op op, int c0, int r1)
    {
        op.from = std::max(op.from, c0); op.upto = std::min(op.upto, r1);
        if (team) {
            team->log(op);
            op_rows(T, ldt, op, c0, std::min(op.from, n), 1);
            op_cols(T, ldt, op, op.upto, r1, w.data());
            if (z0) op_cols(z0, 1, op, 0, 1, w.data());
        } else {
            op_rows(T, ldt, op, c0, n, 1);
            op_cols(T, ldt, op, 0, r1, w.data());
            op_cols(Z, ldz, op, 0, n, w.data());
            if (z0) op_cols(z0, 1, op, 0, 1, w.data());
        }
    }
    // as `emit` with a team, but the calling thread only takes the columns [c0, near) of its share from
    // the left now; the caller owes T(k:k+nr, near:from) itself
    inline void emit_near(Op const &op, int c0, int near, int r1)
    {
        team->log(op);
        op_rows(T, ldt, op, c0, near, 1);
        op_cols(T, ldt, op, 0, r1, w.data());
        if (z0) op_cols(z0, 1, op, 0, 1, w.data());
    }
    inline void publish() { if (team) team->publish(); }
    inline void whole_t() { if (team) team->wait_t(); }
    inline void whole() { if (team) team->wait_all(); }
};

}} // namespace sn::host
