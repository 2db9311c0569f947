// Host-side sequential kernels of the Schur path: the small dense problems that sit on
// the critical path of the multi-shift QR iteration (rows S4/S6/S8 of SURVEY.md 8a).
//
// The reference runs these on a CPU worker through LAPACK (dhseqr, dtrexc, dgehrd,
// dormhr, dlanv2: schur/cpu_utils.c:2248-2309, :2837-3046, :3377-3416, :3493-3594).
// There is no LAPACK here; the published algorithms are written out for the sizes this
// path needs (windows of a few hundred rows).  Everything works on HOST copies of small
// diagonal windows; the bulk of the flops (off-diagonal updates) stays on the GPU.
#include "schur_host.h"
#include "tuning.h"
#include <cmath>
#include <cfloat>
#include <cstring>
#include <vector>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <atomic>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <immintrin.h>
#include "schur_host_team.h"

namespace sn { namespace host {

// ---- helper threads of the sequential window kernels (schur_host_team.h) -----------------------
namespace {
// (per calling thread: every device thread of the in-process multi-GPU path reduces its own replica)
Team &team() { static thread_local Team t; return t; }
thread_local bool g_helpers_on = false;
} // namespace

// Opens / closes a helper session (schur_device brackets a reduction with it when the node has
// the cores; tests switch it on to compare against the serial kernels bit for bit).
void helper_session(bool on, int count)
{
    if (on == g_helpers_on) return;
    if (on) team().open(count); else team().close();
    g_helpers_on = on;
}
static inline Team *team_for(int n) { return (g_helpers_on && n >= 48) ? &team() : nullptr; }

static inline double sign(double a, double b) { return b >= 0.0 ? std::fabs(a) : -std::fabs(a); }

// ---- 2x2 standardisation (LAPACK dlanv2) ---------------------------------------
void lanv2(double &a, double &b, double &c, double &d,
    double &rt1r, double &rt1i, double &rt2r, double &rt2i, double &cs, double &sn)
{
    const double eps = DBL_EPSILON;
    if (c == 0.0) { cs = 1.0; sn = 0.0; }
    else if (b == 0.0) {
        cs = 0.0; sn = 1.0;
        std::swap(a, d); b = -c; c = 0.0;
    }
    else if (a - d == 0.0 && sign(1.0, b) != sign(1.0, c)) { cs = 1.0; sn = 0.0; }
    else {
        double temp = a - d, p = 0.5 * temp;
        double bcmax = std::max(std::fabs(b), std::fabs(c));
        double bcmis = std::min(std::fabs(b), std::fabs(c)) * sign(1.0, b) * sign(1.0, c);
        double scale = std::max(std::fabs(p), bcmax);
        double z = (p / scale) * p + (bcmax / scale) * bcmis;
        if (z >= 4.0 * eps) {                       // real eigenvalues
            z = p + sign(std::sqrt(scale) * std::sqrt(z), p);
            a = d + z; d = d - (bcmax / z) * bcmis;
            double tau = std::hypot(c, z);
            cs = z / tau; sn = c / tau; b = b - c; c = 0.0;
        } else {                                    // complex or nearly equal real
            double sigma = b + c, tau = std::hypot(sigma, temp);
            cs = std::sqrt(0.5 * (1.0 + std::fabs(sigma) / tau));
            sn = -(p / (tau * cs)) * sign(1.0, sigma);
            double aa = a * cs + b * sn, bb = -a * sn + b * cs;
            double cc = c * cs + d * sn, dd = -c * sn + d * cs;
            a = aa * cs + cc * sn; b = bb * cs + dd * sn;
            c = -aa * sn + cc * cs; d = -bb * sn + dd * cs;
            temp = 0.5 * (a + d); a = temp; d = temp;
            if (c != 0.0) {
                if (b != 0.0) {
                    if (sign(1.0, b) == sign(1.0, c)) {
                        double sab = std::sqrt(std::fabs(b)), sac = std::sqrt(std::fabs(c));
                        p = sign(sab * sac, c);
                        tau = 1.0 / std::sqrt(std::fabs(b + c));
                        a = temp + p; d = temp - p; b = b - c; c = 0.0;
                        double cs1 = sab * tau, sn1 = sac * tau;
                        temp = cs * cs1 - sn * sn1; sn = cs * sn1 + sn * cs1; cs = temp;
                    }
                } else { b = -c; c = 0.0; temp = cs; cs = -sn; sn = temp; }
            }
        }
    }
    rt1r = a; rt2r = d;
    if (c == 0.0) { rt1i = rt2i = 0.0; }
    else { rt1i = std::sqrt(std::fabs(b)) * std::sqrt(std::fabs(c)); rt2i = -rt1i; }
}

// Householder vector for x (length n, n <= 4): x <- [beta; v(1:)], returns tau (v(0) = 1).
static double house(int n, double *x)
{
    // the norms by plain sums of squares when every entry is far from the ends of the exponent range
    // (always, in practice), by hypot otherwise: hypot is several times the cost of the whole 3x3 case
    double big = 0.0, small = DBL_MAX, ssq = 0.0;
    for (int i = 1; i < n; i++) { double a = std::fabs(x[i]); big = std::max(big, a); if (a != 0.0) small = std::min(small, a); ssq += a * a; }
    if (big == 0.0) return 0.0;
    double const a0 = std::fabs(x[0]);
    big = std::max(big, a0); if (a0 != 0.0) small = std::min(small, a0);
    double xnorm, nrm;
    if (big < 1e140 && small > 1e-140) { xnorm = std::sqrt(ssq); nrm = std::sqrt(ssq + a0 * a0); }
    else {
        xnorm = 0.0;
        for (int i = 1; i < n; i++) xnorm = std::hypot(xnorm, x[i]);
        nrm = std::hypot(x[0], xnorm);
    }
    if (xnorm == 0.0) return 0.0;
    double alpha = x[0];
    double beta = -sign(nrm, alpha);
    double tau = (beta - alpha) / beta, s = 1.0 / (alpha - beta);
    for (int i = 1; i < n; i++) x[i] *= s;
    x[0] = beta;
    return tau;
}

#define T_(i, j) T[(size_t)(j) * ldt + (i)]
#define Z_(i, j) Z[(size_t)(j) * ldz + (i)]

// ---- double-shift QR on a small Hessenberg matrix (LAPACK dlahqr, full Schur form) ---
// With a team (ap.team): Z, the columns right of the active block and the rows above it are the
// helpers' (see schur_host_team.h); on return T is whole again, Z may still be in flight.
static int small_schur(Applier &ap, double *wr, double *wi)
{
    int const n = ap.n, ldt = ap.ldt; double *const T = ap.T;
    const double ulp = DBL_EPSILON, safmin = DBL_MIN;
    const double smlnum = safmin * ((double)n / ulp);
    const int itmax = 30 * std::max(10, n), kexsh = 10;
    if (n == 0) return 0;
    for (int j = 0; j + 2 < n; j++) { T_(j + 2, j) = 0.0; if (j + 3 < n) T_(j + 3, j) = 0.0; }
    Team *const tm = ap.team;
    constexpr int LOOK = 16;        // measured flat between 8 and 24 (nw = 256, EPYC 9575F)
    auto chunk_at = [&](int k) { return k / LOOK; };         // chunk c: the reflectors at rows [c LOOK, (c+1) LOOK) ...
    auto bnd = [&](int c) { return c * LOOK + 2; };          // ... and the columns [bnd(c), bnd(c+1)): a reflector reaches two columns on
    std::vector<unsigned> far_idx;      // per column: the log position after the last reflector whose helpers' share reaches it
    Op held[LOOK]; int nheld = 0;   // the reflectors of the current chunk
    if (tm) far_idx.assign(n, tm->pending);
    int i = n - 1;
    while (i >= 0) {
        int l = 0;
        bool done = false;
        for (int its = 0; its <= itmax; its++) {
            int k;
            for (k = i; k > l; k--) {
                double sub = std::fabs(T_(k, k - 1));
                if (sub <= smlnum) break;
                double tst = std::fabs(T_(k - 1, k - 1)) + std::fabs(T_(k, k));
                if (tst == 0.0) {
                    if (k - 2 >= 0) tst += std::fabs(T_(k - 1, k - 2));
                    if (k + 1 < n) tst += std::fabs(T_(k + 1, k));
                }
                if (sub <= ulp * tst) {
                    double up = std::fabs(T_(k - 1, k));
                    double ab = std::max(sub, up), ba = std::min(sub, up);
                    double dd = std::fabs(T_(k - 1, k - 1) - T_(k, k));
                    double aa = std::max(std::fabs(T_(k, k)), dd), bb = std::min(std::fabs(T_(k, k)), dd);
                    double s = aa + ab;
                    if (ba * (ab / s) <= std::max(smlnum, ulp * (bb * (aa / s)))) break;
                }
            }
            l = k;
            if (l > 0) T_(l, l - 1) = 0.0;
            if (l >= i - 1) { done = true; break; }

            double h11, h21, h12, h22;
            if (its > 0 && its % (2 * kexsh) == 0) {
                double s = std::fabs(T_(l + 1, l)) + std::fabs(T_(l + 2, l + 1));
                h11 = 0.75 * s + T_(l, l); h12 = -0.4375 * s; h21 = s; h22 = h11;
            } else if (its > 0 && its % kexsh == 0) {
                double s = std::fabs(T_(i, i - 1)) + std::fabs(T_(i - 1, i - 2));
                h11 = 0.75 * s + T_(i, i); h12 = -0.4375 * s; h21 = s; h22 = h11;
            } else {
                h11 = T_(i - 1, i - 1); h21 = T_(i, i - 1); h12 = T_(i - 1, i); h22 = T_(i, i);
            }
            double rt1r, rt1i, rt2r, rt2i;
            double s = std::fabs(h11) + std::fabs(h12) + std::fabs(h21) + std::fabs(h22);
            if (s == 0.0) rt1r = rt1i = rt2r = rt2i = 0.0;
            else {
                h11 /= s; h21 /= s; h12 /= s; h22 /= s;
                double tr = 0.5 * (h11 + h22);
                double det = (h11 - tr) * (h22 - tr) - h12 * h21;
                double rtdisc = std::sqrt(std::fabs(det));
                if (det >= 0.0) { rt1r = tr * s; rt2r = rt1r; rt1i = rtdisc * s; rt2i = -rt1i; }
                else {
                    rt1r = tr + rtdisc; rt2r = tr - rtdisc;
                    if (std::fabs(rt1r - h22) <= std::fabs(rt2r - h22)) { rt1r *= s; rt2r = rt1r; }
                    else { rt2r *= s; rt1r = rt2r; }
                    rt1i = rt2i = 0.0;
                }
            }
            double v[3];
            int m;
            for (m = i - 2; m >= l; m--) {
                double h21s = std::fabs(T_(m + 1, m));
                double ss = std::fabs(T_(m, m) - rt2r) + std::fabs(rt2i) + h21s;
                h21s = T_(m + 1, m) / ss;
                v[0] = h21s * T_(m, m + 1) + (T_(m, m) - rt1r) * ((T_(m, m) - rt2r) / ss) - rt1i * (rt2i / ss);
                v[1] = h21s * (T_(m, m) + T_(m + 1, m + 1) - rt1r - rt2r);
                v[2] = h21s * T_(m + 2, m + 1);
                ss = std::fabs(v[0]) + std::fabs(v[1]) + std::fabs(v[2]);
                v[0] /= ss; v[1] /= ss; v[2] /= ss;
                if (m == l) break;
                double nb = std::fabs(T_(m - 1, m - 1)) + std::fabs(T_(m, m)) + std::fabs(T_(m + 1, m + 1));
                if (std::fabs(T_(m, m - 1)) * (std::fabs(v[1]) + std::fabs(v[2])) <= ulp * std::fabs(v[0]) * nb)
                    break;
            }
            // With a team the sweep also leaves the FAR columns of the active block to the helpers.  The
            // columns are cut into chunks of LOOK at fixed positions; while the bulge is in chunk c the
            // chain applies its reflectors at once to the columns up to the end of chunk c, LATER -- in
            // one go, when the bulge enters chunk c+1 -- to the columns of chunk c+1, and not at all to
            // the columns beyond, which the helpers update (Op::from) and hand back a chunk at a time:
            // a chunk comes home when the bulge enters it, and what the helpers still owe it then was
            // published a whole chunk of the chain's work earlier.
            int c = chunk_at(m), imm = i + 1, lz = i + 1;
            if (tm) { imm = std::min(bnd(c + 1), i + 1); lz = std::min(bnd(c + 2), i + 1); }
            auto next_chunk = [&]() {
                tm->publish();
                unsigned const owed = lz > imm ? far_idx[lz - 1] : 0;       // before this chunk's own marks
                for (int x = lz; x < n; x++) far_idx[x] = tm->pending;
                if (lz > imm) {
                    tm->wait_t_index(owed);
                    for (int q = 0; q < nheld; q++) op_rows(T, ldt, held[q], imm, lz, 1);
                }
                nheld = 0;
                c++;
                imm = std::min(bnd(c + 1), i + 1); lz = std::min(bnd(c + 2), i + 1);
            };
            for (int k2 = m; k2 <= i - 1; k2++) {
                int nr = std::min(3, i - k2 + 1);
                if (tm && k2 >= (c + 1) * LOOK) next_chunk();
                if (k2 > m) { v[0] = T_(k2, k2 - 1); v[1] = T_(k2 + 1, k2 - 1); if (nr == 3) v[2] = T_(k2 + 2, k2 - 1); }
                double t1 = house(nr, v);
                if (k2 > m) { T_(k2, k2 - 1) = v[0]; T_(k2 + 1, k2 - 1) = 0.0; if (k2 < i - 1) T_(k2 + 2, k2 - 1) = 0.0; }
                else if (m > l) T_(k2, k2 - 1) *= (1.0 - t1);
                // the chain's share: the rows inside the active block from the left, all of the columns.
                // (The rows above l are not read again either, but they become "columns right of the
                // active block" of a later, higher block: T(r, c) would then be one helper's as a row
                // above and another's as a column to the right, and the order between the two is lost.)
                Op const op{k2, (short)nr, 0, lz, 0, v[1], nr == 3 ? v[2] : 0.0, 0.0, t1};
                if (tm) { held[nheld++] = op; ap.emit_near(op, k2, imm, std::min(k2 + 3, i) + 1); }
                else ap.emit(op, k2, std::min(k2 + 3, i) + 1);
            }
            if (tm) {
                next_chunk();                       // the last chunk's later columns (column i can be one)
                tm->wait_t_index(far_idx[i]);       // every column of the active block is whole again
            }
        }
        if (!done) { ap.whole_t(); return i + 1; }
        if (l == i) { wr[i] = T_(i, i); wi[i] = 0.0; }
        else {
            double cs, sn;
            lanv2(T_(i - 1, i - 1), T_(i - 1, i), T_(i, i - 1), T_(i, i),
                wr[i - 1], wi[i - 1], wr[i], wi[i], cs, sn);
            // the 2x2 block itself is standardised by lanv2: columns right of it, rows above it
            ap.emit(Op{i - 1, 2, 1, i + 1, 0, cs, sn, 0.0, 0.0}, i + 1, i - 1);
            if (tm) { tm->publish(); for (int x = i + 1; x < n; x++) far_idx[x] = tm->pending; }
        }
        i = l - 1;
    }
    ap.whole_t();
    return 0;
}

int small_schur(int n, double *T, int ldt, double *Z, int ldz, double *wr, double *wi)
{
    if (n == 0) return 0;
    Applier ap{T, ldt, Z, ldz, n, team_for(n)};
    ap.w.resize(n);
    if (ap.team) ap.team->begin(T, ldt, Z, ldz, n);
    int const info = small_schur(ap, wr, wi);
    ap.whole();
    return info;
}

// ---- swapping adjacent diagonal blocks of a real Schur form (LAPACK dlaexc) ----------
// Blocks T11 (n1 x n1) at j1 and T22 (n2 x n2) at j1+n1, n1,n2 in {1,2}.  Returns 0 if
// swapped, 1 if the swap was rejected as too inaccurate (T, Z untouched).
// `hi`, `lo`: the caller reads nothing right of column hi or above row lo before the next wait on the
// team (Applier::whole_t): those parts of the update, and Z, may be left to the helpers.
static int swap_blocks(Applier &ap, int j1, int n1, int n2, int hi, int lo)
{
    int const n = ap.n, ldt = ap.ldt; double *const T = ap.T;
    if (n1 == 0 || n2 == 0) return 0;
    if (j1 + n1 >= n) return 0;
    const int nd = n1 + n2;
    if (n1 == 1 && n2 == 1) {
        double t11 = T_(j1, j1), t22 = T_(j1 + 1, j1 + 1);
        double f = T_(j1, j1 + 1), g = t22 - t11;
        double r = std::hypot(f, g), cs, sn;
        if (r == 0.0) { cs = 1.0; sn = 0.0; } else { cs = f / r; sn = g / r; }
        ap.emit(Op{j1, 2, 1, hi, lo, cs, sn, 0.0, 0.0}, j1 + 2, j1);
        T_(j1, j1) = t22; T_(j1 + 1, j1 + 1) = t11;
        return 0;
    }
    // local copy D of the nd x nd diagonal block
    double D[4][4];
    double dnorm = 0.0;
    for (int i = 0; i < nd; i++)
        for (int j = 0; j < nd; j++) { D[i][j] = T_(j1 + i, j1 + j); dnorm = std::max(dnorm, std::fabs(D[i][j])); }
    const double eps = DBL_EPSILON, smlnum = DBL_MIN / eps;
    const double thresh = std::max(10.0 * eps * dnorm, smlnum);

    // Sylvester equation T11 X - X T22 = T12 as a (n1*n2) linear system, complete pivoting
    const int ns = n1 * n2;
    double K[4][4] = {{0}}, rhs[4], X[2][2] = {{0}};
    for (int j = 0; j < n2; j++)
        for (int i = 0; i < n1; i++) {
            int row = j * n1 + i;
            rhs[row] = D[i][n1 + j];
            for (int p = 0; p < n1; p++) K[row][j * n1 + p] += D[i][p];
            for (int q = 0; q < n2; q++) K[row][q * n1 + i] -= D[n1 + q][n1 + j];
        }
    {
        int perm[4] = {0, 1, 2, 3};
        double kmax = 0.0;
        for (int i = 0; i < ns; i++) for (int j = 0; j < ns; j++) kmax = std::max(kmax, std::fabs(K[i][j]));
        double smin = std::max(eps * kmax, smlnum);
        for (int c = 0; c < ns; c++) {
            int pi = c, pj = c; double pv = 0.0;
            for (int i = c; i < ns; i++) for (int j = c; j < ns; j++)
                if (std::fabs(K[i][j]) > pv) { pv = std::fabs(K[i][j]); pi = i; pj = j; }
            if (pi != c) { for (int j = 0; j < ns; j++) std::swap(K[pi][j], K[c][j]); std::swap(rhs[pi], rhs[c]); }
            if (pj != c) { for (int i = 0; i < ns; i++) std::swap(K[i][pj], K[i][c]); std::swap(perm[pj], perm[c]); }
            if (std::fabs(K[c][c]) < smin) K[c][c] = smin;
            for (int i = c + 1; i < ns; i++) {
                double f = K[i][c] / K[c][c];
                for (int j = c; j < ns; j++) K[i][j] -= f * K[c][j];
                rhs[i] -= f * rhs[c];
            }
        }
        double sol[4];
        for (int c = ns - 1; c >= 0; c--) {
            double s = rhs[c];
            for (int j = c + 1; j < ns; j++) s -= K[c][j] * sol[j];
            sol[c] = s / K[c][c];
        }
        for (int c = 0; c < ns; c++) { int col = perm[c]; X[col % n1][col / n1] = sol[c]; }
    }
    // QR of M = [-X; I] (nd x n2): Q^T [T11 T12; 0 T22] Q = [T22' *; 0 T11']
    double M[4][2], vv[2][4], tau[2];
    for (int j = 0; j < n2; j++) {
        for (int i = 0; i < n1; i++) M[i][j] = -X[i][j];
        for (int i = 0; i < n2; i++) M[n1 + i][j] = (i == j) ? 1.0 : 0.0;
    }
    for (int k = 0; k < n2; k++) {
        double x[4]; int len = nd - k;
        for (int i = 0; i < len; i++) x[i] = M[k + i][k];
        tau[k] = house(len, x);
        vv[k][0] = 1.0; for (int i = 1; i < len; i++) vv[k][i] = x[i];
        for (int j = k + 1; j < n2; j++) {
            double s = 0.0;
            for (int i = 0; i < len; i++) s += vv[k][i] * M[k + i][j];
            s *= tau[k];
            for (int i = 0; i < len; i++) M[k + i][j] -= s * vv[k][i];
        }
    }
    // trial on the local copy
    for (int k = 0; k < n2; k++) {
        int len = nd - k;
        for (int j = 0; j < nd; j++) {          // left
            double s = 0.0;
            for (int i = 0; i < len; i++) s += vv[k][i] * D[k + i][j];
            s *= tau[k];
            for (int i = 0; i < len; i++) D[k + i][j] -= s * vv[k][i];
        }
        for (int i = 0; i < nd; i++) {          // right
            double s = 0.0;
            for (int j = 0; j < len; j++) s += D[i][k + j] * vv[k][j];
            s *= tau[k];
            for (int j = 0; j < len; j++) D[i][k + j] -= s * vv[k][j];
        }
    }
    double low = 0.0;
    for (int i = n2; i < nd; i++) for (int j = 0; j < n2; j++) low = std::max(low, std::fabs(D[i][j]));
    if (low > thresh) return 1;

    // accept: apply to T and Z
    for (int k = 0; k < n2; k++) {
        int const len = nd - k;
        ap.emit(Op{j1 + k, (short)len, 0, hi, lo, vv[k][1], len > 2 ? vv[k][2] : 0.0, len > 3 ? vv[k][3] : 0.0, tau[k]}, j1, j1 + nd);
    }
    for (int i = n2; i < nd; i++) for (int j = 0; j < n2; j++) T_(j1 + i, j1 + j) = 0.0;
    // standardise the new 2x2 blocks
    auto standardise = [&](int p) {
        double rt1r, rt1i, rt2r, rt2i, cs, sn;
        lanv2(T_(p, p), T_(p, p + 1), T_(p + 1, p), T_(p + 1, p + 1), rt1r, rt1i, rt2r, rt2i, cs, sn);
        ap.emit(Op{p, 2, 1, hi, lo, cs, sn, 0.0, 0.0}, p + 2, p);
    };
    if (n2 == 2) standardise(j1);
    if (n1 == 2) standardise(j1 + n2);
    return 0;
}

// Moves the diagonal block starting at row `from` up to row `to` (to <= from) by adjacent
// swaps (LAPACK dtrexc, upward direction only; schur/cpu_utils.c:3377-3416).  Returns the
// row where the block ended up (== to unless a swap was rejected).
static int move_block_up(Applier &ap, int from, int to, int hi, int lo)
{
    int const n = ap.n, ldt = ap.ldt; double *const T = ap.T;
    int here = from;
    int nbf = (here + 1 < n && T_(here + 1, here) != 0.0) ? 2 : 1;
    while (here > to) {
        int nbabove = (here - 2 >= 0 && T_(here - 1, here - 2) != 0.0) ? 2 : 1;
        if (here - nbabove < to) break;       // `to` points into the middle of a block
        int j1 = here - nbabove;
        if (swap_blocks(ap, j1, nbabove, nbf, hi, lo) != 0) break;
        here = j1;
        if (nbf == 2 && T_(here + 1, here) == 0.0) {
            // the moving 2x2 block split into two 1x1 blocks: move them one at a time
            int a = move_block_up(ap, here, to, hi, lo);
            if (a != to) return a;
            move_block_up(ap, here + 1, to + 1, hi, lo);
            return a;
        }
    }
    ap.publish();
    return here;
}
int move_block_up(int n, double *T, int ldt, double *Z, int ldz, int from, int to)
{
    Applier ap{T, ldt, Z, ldz, n};
    return move_block_up(ap, from, to, n, 0);
}

// ---- eigenvalue reordering inside one diagonal window (reorder/cpu.c reorder_window; LAPACK
// dtrsen's loop of dtrexc calls) --------------------------------------------------------------
// T (w x w, quasi-triangular) <- Z^T T Z with the selected diagonal blocks moved to the top of
// the window, in their original order; Z accumulates from the right.  sel[i] != 0 marks the rows
// of selected blocks (both rows of a 2x2 block carry the mark of the block: either one set
// selects it).  On return sel holds the marks of the rows in their NEW order.  Returns the number
// of rows of selected blocks now at the top; *failed is set when a swap was rejected (the blocks
// involved are too close for a stable exchange): the blocks moved so far stay where they are, the
// rejected one and everything selected below it keep their marks at their current rows.
int reorder_window(int w, double *T, int ldt, double *Z, int ldz, int *sel, int *failed)
{
    *failed = 0;
    int top = 0, i = 0;
    while (i < w) {
        int const bs = (i + 1 < w && T_(i + 1, i) != 0.0) ? 2 : 1;
        bool const marked = sel[i] != 0 || (bs == 2 && sel[i + 1] != 0);
        if (!marked) { i += bs; continue; }
        if (i > top) {
            int const at = move_block_up(w, T, ldt, Z, ldz, i, top);
            // marks: rows [top, at) are unselected blocks that stayed above; the moved block sits
            // at [at, at + bs); the rows it passed moved down by bs
            for (int r = i + bs - 1; r >= at + bs; r--) sel[r] = 0;       // rows that moved down
            for (int r = at; r < at + bs; r++) sel[r] = 1;
            if (at != top) {                    // swap rejected on the way up
                *failed = 1;
                return top;
            }
        } else for (int r = i; r < i + bs; r++) sel[r] = 1;
        top += bs;
        i += bs;
    }
    return top;
}

// ---- shifts (schur/cpu_utils.c:3493-3594) ---------------------------------------------
void extract_eigenvalues(int n, const double *T, int ldt, double *wr, double *wi)
{
    for (int i = 0; i < n; i++) {
        if (i + 1 < n && T_(i + 1, i) != 0.0) {
            double a = T_(i, i), b = T_(i, i + 1), c = T_(i + 1, i), d = T_(i + 1, i + 1), cs, sn;
            lanv2(a, b, c, d, wr[i], wi[i], wr[i + 1], wi[i + 1], cs, sn);
            i++;
        } else { wr[i] = T_(i, i); wi[i] = 0.0; }
    }
}

int extract_shifts(int n, const double *T, int ldt, double *wr, double *wi)
{
    extract_eigenvalues(n, T, ldt, wr, wi);
    return order_shifts(n, wr, wi);
}

// ordering / pairing of a list of eigenvalues as shifts (schur/cpu_utils.c:3522-3594)
int order_shifts(int n, double *wr, double *wi)
{
    // zero / non-finite shifts go to the end and are dropped (cpu_utils.c:3529-3552)
    int end = n;
    for (int i = end - 1; i >= 0; i--) {
        bool bad = (wr[i] == 0.0 && wi[i] == 0.0) || !std::isfinite(wr[i]) || !std::isfinite(wi[i]);
        if (bad) { std::swap(wr[i], wr[end - 1]); std::swap(wi[i], wi[end - 1]); end--; }
    }
    // ascending |re|+|im| (stable, like the reference's bubble sort :3555-3577)
    std::vector<int> idx(end);
    for (int i = 0; i < end; i++) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
        return std::fabs(wr[a]) + std::fabs(wi[a]) < std::fabs(wr[b]) + std::fabs(wi[b]); });
    std::vector<double> r(end), im(end);
    for (int i = 0; i < end; i++) { r[i] = wr[idx[i]]; im[i] = wi[idx[i]]; }
    for (int i = 0; i < end; i++) { wr[i] = r[i]; wi[i] = im[i]; }
    // pair up: (real, real) or (complex, conjugate)  (:3581-3597)
    for (int i = 0; i + 2 < end; i += 2) {
        if (wi[i] != -wi[i + 1]) {
            double sr = wr[i], si = wi[i];
            wr[i] = wr[i + 1]; wr[i + 1] = wr[i + 2]; wr[i + 2] = sr;
            wi[i] = wi[i + 1]; wi[i + 1] = wi[i + 2]; wi[i + 2] = si;
        }
    }
    return end;
}

// ---- Hessenberg reduction of the leading ns x ns part of a window (dgehd2 + dormhr) ----
// T is the nw x nw window (spike already embedded by the caller in the column to the left,
// handled through `spike`): rows/cols [0, ns) are reduced, the reflectors are applied to
// T(0:ns, ns:nw) from the left and accumulated into Z(:, 0:ns) from the right.
// `arena` receives the reflector vectors (the helpers read them until the next wait on the team).
static void hessenberg_small(Applier &ap, int ns, double *arena)
{
    int const ldt = ap.ldt; double *const T = ap.T;
    for (int k = 0; k + 2 < ns; k++) {
        int len = ns - k - 1;
        double *v = arena; arena += len;
        for (int i = 0; i < len; i++) v[i] = T_(k + 1 + i, k);
        double tau = house(len, v);
        T_(k + 1, k) = v[0];
        for (int i = 1; i < len; i++) T_(k + 1 + i, k) = 0.0;
        if (tau == 0.0) continue;
        v[0] = 1.0;
        // left: rows k+1..ns-1, columns k+1..nw-1 (right of ns: the helpers'); right: columns k+1..ns-1,
        // rows 0..ns-1 (the rows above k+2 are not read again: the helpers'); Z
        Op op{k + 1, (short)len, 2, ns, k + 2, 0.0, 0.0, 0.0, tau};
        op_set_vector(op, v);
        ap.emit(op, k + 1, ns);
        if ((k & 3) == 3) ap.publish();
    }
    ap.publish();
}

// ---- one deflation window of the blocked AED (schur/cpu.c:638-1006 starneig_cpu_deflate,
// driven by schur/core.c:1070-1252) ----------------------------------------------------------
// T: w x w quasi-triangular diagonal window of the (already Schur-reduced) AED window; the
// bottom `carried` rows hold blocks that an earlier deflation window found undeflatable, the
// rows above them are unchecked.  spike[0:w] is the window's segment of the spike row
// sub * Z_aed(0,:).  The carried blocks are moved to the top of the window, then the unchecked
// blocks are tested from the bottom (LAPACK dlaqr3 order: a block is tested when it is the last
// undeflated one; an undeflatable block joins the carried ones at the top).  On return
// [0, *undeflated) holds the undeflatable blocks, [*undeflated, w) the deflated ones, Z the
// accumulated swaps and spike <- spike * Z.  Returns 0, or 1 if a swap was rejected (everything
// above the rejected block counts as undeflatable then).
int deflate_window(int w, double *T, int ldt, double *Z, int ldz, double *spike, double sub,
    double thres, int carried, int *undeflated)
{
    int rc = 0;
    // (1) carried blocks to the top, in order
    int top = 0;
    for (int i = w - carried; i < w;) {
        int const bs = (i + 1 < w && T_(i + 1, i) != 0.0) ? 2 : 1;
        if (i > top) {
            int const at = move_block_up(w, T, ldt, Z, ldz, i, top);
            if (at != top) { *undeflated = w; rc = 1; goto done; }   // nothing can be tested behind it
        }
        top += bs; i += bs;
    }
    {
        // (2) test the unchecked blocks, now in [top, w), from the bottom
        const double ulp = DBL_EPSILON, smlnum = DBL_MIN * ((double)w / ulp);
        auto cur = [&](int col) { double v = 0.0; for (int k = 0; k < w; k++) v += spike[k] * Z_(k, col); return v; };
        int i = w - 1;
        while (top <= i) {
            bool const two = (top <= i - 1 && T_(i, i - 1) != 0.0);
            double sp = std::fabs(cur(i));
            if (two) sp = std::max(sp, std::fabs(cur(i - 1)));
            bool deflatable;
            if (thres > 0.0) deflatable = sp < thres;
            else {
                double foo = std::fabs(T_(i, i));
                if (two) foo += std::sqrt(std::fabs(T_(i, i - 1))) * std::sqrt(std::fabs(T_(i - 1, i)));
                if (foo == 0.0) foo = std::fabs(sub);
                deflatable = sp < std::max(smlnum, ulp * foo);
            }
            int const bs = two ? 2 : 1;
            if (deflatable) i -= bs;
            else {
                int const at = move_block_up(w, T, ldt, Z, ldz, i - bs + 1, top);
                if (at != top) { top = i + 1; rc = 1; break; }
                top += bs;
            }
        }
        *undeflated = top;
    }
done:
    {
        std::vector<double> ns(w);
        for (int j = 0; j < w; j++) { double v = 0.0; for (int k = 0; k < w; k++) v += spike[k] * Z_(k, j); ns[j] = v; }
        for (int j = 0; j < w; j++) spike[j] = ns[j];
    }
    return rc;
}

// ---- aggressive early deflation on a host window (schur/cpu_utils.c:2837-3046) ---------
// T: nw x nw Hessenberg window; `sub` = the sub-diagonal entry that couples the window to
// the matrix above it (0 if the window starts the active block).  On return T holds
// [ Hessenberg (ns x ns) | * ; 0 | Schur (nd x nd) ], Z the accumulated transformation,
// spike[0:nw] the new first column below the coupling row (sub * Z(0,:), Hessenberg part
// compressed to its first entry), shifts in sr/si.  Returns nd (deflated eigenvalues).
AedResult aed_window(int nw, double *T, int ldt, double *Z, int ldz, double sub,
    double thres, double *spike, double *sr, double *si)
{
    AedResult res{0, 0, 0};
    bool const prof = tuning().aed_profile;
    static double t_schur = 0, t_reorder = 0, t_hess = 0; static int calls = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
#ifdef SN_TEST_HOOKS
    // SN_AED_DUMP=<file> (test library only): every 6th window as it arrives, for replay off the GPU
    // box (scratch/aed_replay.py).  Record: int32 nw, float64 sub, thres, then nw*nw column-major.
    if (char const *path = getenv("SN_AED_DUMP")) {
        static int seen = 0;
        if (seen++ % 6 == 0 && seen < 6 * 48) {
            if (FILE *f = fopen(path, "ab")) {
                int32_t hdr = nw; fwrite(&hdr, 4, 1, f); fwrite(&sub, 8, 1, f); fwrite(&thres, 8, 1, f);
                for (int j = 0; j < nw; j++) fwrite(&T_(0, j), 8, nw, f);
                fclose(f);
            }
        }
    }
#endif
    double t0 = now();
    for (int j = 0; j < nw; j++) for (int i = 0; i < nw; i++) Z_(i, j) = (i == j) ? 1.0 : 0.0;
    std::vector<double> wr(nw), wi(nw), z0(nw, 0.0);
    z0[0] = 1.0;
    // Z(0, :) -- all the deflation test reads of Z -- is kept on this thread; Z itself, and the parts of
    // T the three phases below do not read again, may be behind (schur_host_team.h) until `whole`
    Applier ap{T, ldt, Z, ldz, nw, team_for(nw), z0.data()};
    ap.w.resize(nw);
    struct Whole { Applier &a; ~Whole() { a.whole(); } } whole_on_return{ap};
    if (ap.team) ap.team->begin(T, ldt, Z, ldz, nw);
    int info = small_schur(ap, wr.data(), wi.data());
    int roof = 0;
    if (info != 0) {
        // rows [0, info) did not converge: only the trailing part is in Schur form
        roof = info;
        res.failed = 1;
    }
    double t1 = now();
    const double ulp = DBL_EPSILON, smlnum = DBL_MIN * ((double)nw / ulp);
    int top = roof;           // undeflatable blocks accumulate in [roof, top)
    int i = nw - 1;
    while (top <= i) {
        bool two = (top <= i - 1 && T_(i, i - 1) != 0.0);
        bool deflatable;
        if (thres > 0.0) {      // norm-stable criterion (cpu_utils.c:2891-2931)
            deflatable = std::fabs(sub * z0[i]) < thres &&
                (!two || std::fabs(sub * z0[i - 1]) < thres);
        } else {                // LAPACK-style criterion (:2937-2988)
            double foo = std::fabs(T_(i, i));
            if (two) foo += std::sqrt(std::fabs(T_(i, i - 1))) * std::sqrt(std::fabs(T_(i - 1, i)));
            if (foo == 0.0) foo = std::fabs(sub);
            double sp = std::fabs(sub * z0[i]);
            if (two) sp = std::max(sp, std::fabs(sub * z0[i - 1]));
            deflatable = sp < std::max(smlnum, ulp * foo);
        }
        int bs = two ? 2 : 1;
        if (deflatable) i -= bs;
        else {
            // on its way up the block only meets blocks in [top, i]: the columns right of i and the rows
            // above top take no part in any later swap or test
            int from = i - bs + 1;
            int at = move_block_up(ap, from, top, i + 1, top);
            if (at != top) { top = i + 1; break; }   // swap rejected: nothing below `i` deflates
            top += bs;
        }
    }
    ap.whole_t();
    double t2 = now();
    res.deflated = nw - top;
    int ns = top;
    // shifts from the undeflated part; "extract something" if it is too small (:3000-3010)
    if (ns - roof >= 2) res.shifts = extract_shifts(ns - roof, &T_(roof, roof), ldt, sr, si);
    else res.shifts = extract_shifts(nw, T, ldt, sr, si);
    // spike = sub * first row of Z
    for (int j = 0; j < nw; j++) spike[j] = sub * z0[j];
    ap.z0 = nullptr;
    if (res.deflated == 0 && roof == 0) return res;     // caller discards T/Z
    for (int j = ns; j < nw; j++) spike[j] = 0.0;        // deflated: below the threshold
    std::vector<double> arena;
    if (ns > 1 && sub != 0.0) {
        // compress the spike to its first entry with one reflector, then restore Hessenberg
        arena.resize((size_t)ns * (ns + 3) / 2 + 8);
        double *v = arena.data();
        for (int j = 0; j < ns; j++) v[j] = spike[j];
        double tau = house(ns, v);
        spike[0] = v[0];
        for (int j = 1; j < ns; j++) spike[j] = 0.0;
        if (tau != 0.0) {
            v[0] = 1.0;
            // left on rows [0,ns) (columns right of ns: the helpers'), right on cols [0,ns), rows [0,ns); Z
            Op op{0, (short)ns, 2, ns, 0, 0.0, 0.0, 0.0, tau};
            op_set_vector(op, v);
            ap.emit(op, 0, ns);
            ap.publish();
        }
        hessenberg_small(ap, ns, v + ns);
    }
    ap.whole();
    if (prof) {
        t_schur += t1 - t0; t_reorder += t2 - t1; t_hess += now() - t2; calls++;
        if (calls % 20 == 0) {
            fprintf(stderr, "[aed] calls %d, the last 20 per call: schur %.2f ms, deflation %.2f ms, hessenberg %.2f ms (nw %d)\n",
                calls, 50.0 * t_schur, 50.0 * t_reorder, 50.0 * t_hess, nw);
            t_schur = t_reorder = t_hess = 0;
        }
    }
    return res;
}

}} // namespace sn::host

#ifdef SN_TEST_HOOKS   // compiled into libstarneig_amd_test.so only (csrc/Makefile), never into the product library
// ---- test hooks (host-only; NOT part of the public C-ABI, used by tests/ on CPU) -------
extern "C" {
__attribute__((visibility("default")))
int sn_internal_small_schur(int n, double *T, int ldt, double *Z, int ldz, double *wr, double *wi)
{ return sn::host::small_schur(n, T, ldt, Z, ldz, wr, wi); }
__attribute__((visibility("default")))
int sn_internal_move_block_up(int n, double *T, int ldt, double *Z, int ldz, int from, int to)
{ return sn::host::move_block_up(n, T, ldt, Z, ldz, from, to); }
__attribute__((visibility("default")))
void sn_internal_helper_session(int on) { sn::host::helper_session(on != 0, on > 1 ? on : 5); }
__attribute__((visibility("default")))
int sn_internal_deflate_window(int w, double *T, int ldt, double *Z, int ldz, double *spike, double sub,
    double thres, int carried, int *undeflated)
{ return sn::host::deflate_window(w, T, ldt, Z, ldz, spike, sub, thres, carried, undeflated); }
__attribute__((visibility("default")))
int sn_internal_reorder_window(int w, double *T, int ldt, double *Z, int ldz, int *sel, int *failed)
{ return sn::host::reorder_window(w, T, ldt, Z, ldz, sel, failed); }
__attribute__((visibility("default")))
int sn_internal_extract_shifts(int n, const double *T, int ldt, double *wr, double *wi)
{ return sn::host::extract_shifts(n, T, ldt, wr, wi); }
__attribute__((visibility("default")))
int sn_internal_aed_window(int nw, double *T, int ldt, double *Z, int ldz, double sub,
    double thres, double *spike, double *sr, double *si, int *out3)
{
    sn::host::AedResult r = sn::host::aed_window(nw, T, ldt, Z, ldz, sub, thres, spike, sr, si);
    out3[0] = r.deflated; out3[1] = r.shifts; out3[2] = r.failed;
    return 0;
}
}
#endif  // SN_TEST_HOOKS
