// opensot_amd/csrc/osot_qp_big.h -- the BackEnd-convention QP (BackEnd.h:125-150: H, g, lA <= A x <= uA, l <= x <= u) for 65 .. 128
// variables: what osot_qp_solve_batch and the osot_backend_* plugin surface run when the problem is wider than the 64 lanes of a
// wavefront (a 45-DoF robot, a 38-DoF floating-base inverse-dynamics stack with five 6-D contacts: include/OpenSoT/Task.h has no limit).
//
// One WORKGROUP of 256 threads per QP instead of one wavefront.  Same method as the wavefront core (osot_qp_core.h): Goldfarb-Idnani
// dual active set (the reference's eiQuadProg back-end, external/eiQuadProg-ext/include/eiquadprog.hpp), H + eps I = L L', J = L^-T
// explicit, every addition ONE Householder reflection of J2 (the new column of R is [d1; alpha]), removals by Givens rotations of R and
// the matching columns of J, bilateral rows and box bounds native, equality rows (lA == uA) eliminated first and never dropped.
// Data: J (n x n) and the Cholesky factor live in a per-workgroup slice of device memory (256 KB at n = 128: L2-resident; every pass
// over them is a parallel loop with coalesced or row-private accesses), the packed triangular R of the working set, the iterate and the
// work vectors in LDS.  The n^2 work of an iteration (d = J'n, z = J2 d2, the reflection / the rotations of J, the row scan) is spread
// over the threads; the iq^2 work on R (back substitution, re-triangularisation after a removal) and the scalar decisions run on thread 0
// out of LDS; a workgroup barrier separates the sections.  This is the coverage path for wide problems, not a tuned kernel: see DESIGN.md.
//
// The body is written against a two-member "team" (thread id, thread count, barrier) so that the SAME source compiles for the host
// with a team of one (tests/emu: the parity tests of this file run it on the CPU against the oracle; a section is then a plain loop).
#pragma once
#include <cmath>
#include <cstddef>

#if defined(__HIPCC__) && !defined(OSOT_BIG_HOST)
#define OSOT_BIG_FN __device__ inline
#else
#define OSOT_BIG_FN inline
#endif

namespace osot {
namespace big {

constexpr int kMaxVars = 128;
constexpr int kMaxRows = 2048;
constexpr double kInf = 1.0e20;       // QPOasesBackEnd::checkINFTY clamp (QPOasesBackEnd.cpp:339-356)
constexpr double kViol = 1.0e-11;     // a slack below -kViol max(1, |bound|) counts as violated         (osot_qp_core.h: kViolTol)
constexpr double kEq = 1.0e-9;        // consistency of a linearly dependent equality row                (kEqTol)
constexpr double kDep2 = 1.0e-24;     // |d2|^2 <= kDep2 |d|^2: the normal is in the span of the working set (kDepTol2)
constexpr double kRatio = 1.0e-14;    // dual ratio test: r_k counts as positive above this fraction of max |r| (kRatioTol)
enum { ST_SOLVED = 0, ST_INFEASIBLE = 1, ST_MAX_ITER = 2, ST_NOT_PD = 3 };   // = OSOT_STATUS_*

struct Args {                         // one instance
    int n, nc, max_iter;
    double eps;
    const double *H, *g, *A, *lA, *uA, *l, *u;   // l / u null: no box
    double* x;
    int* status;
    int* iters;                       // may be null
    double* Lw;                       // workspace [n][n]: H + eps I -> its Cholesky factor (lower triangle)
    double* J;                        // workspace [n][n]: J = L^-T, rotated / reflected as the working set changes
};
enum { SI_IQ = 0, SI_ME, SI_IP, SI_SIDE, SI_ACT, SI_L, SI_ST, SI_ITERS, SI_DONE, SI_COUNT = 16 };
enum { SD_T = 0, SD_ND2, SD_DD, SD_ALPHA, SD_BETA, SD_SIP, SD_PIV, SD_BND, SD_COUNT = 16 };
struct Shared {                       // LDS on the device, heap on the host
    double* R;                        // packed upper triangle of the working set's factor: R(i, j) at j (j + 1) / 2 + i, i <= j
    double *x, *np, *d, *z, *r, *e;   // [n] each: iterate, normal, d = J'n, z = J2 d2, r = R^-1 d1, scratch
    double* u;                        // [n + 1] multipliers by position (+ the candidate's)
    double* cval;                     // [n + nc] violation of each inactive constraint (0: none)
    double *cs_c, *cs_s;              // [n] the rotations of a removal
    double* sc;                       // [SD_COUNT] shared scalars (written by thread 0 between two barriers)
    int *aset, *aside;                // [n] working set by position: constraint code (variable k, or n + row), side (+1 lower, -1 upper)
    int *bstate;                      // [n] 0 free, 1 lower active, 2 upper active
    int *rstate;                      // [nc] 0 inactive, 1 lower active, 2 upper active, 3 equality (in the set for good, or dependent and consistent)
    int* cside;                       // [n + nc]
    int* si;                          // [SI_COUNT]
};
inline size_t shared_doubles(int n, int nc) {
    return (size_t)n * (n + 1) / 2 + 6 * (size_t)n + (n + 1) + (size_t)(n + nc) + 2 * (size_t)n + SD_COUNT;
}
inline size_t shared_ints(int n, int nc) { return 3 * (size_t)n + nc + (size_t)(n + nc) + SI_COUNT; }
inline size_t shared_bytes(int n, int nc) { return 8 * ((shared_doubles(n, nc) + 1) & ~(size_t)1) + 4 * shared_ints(n, nc); }
OSOT_BIG_FN Shared carve(void* base, int n, int nc) {
    Shared s;
    double* p = reinterpret_cast<double*>(base);
    s.R = p; p += (size_t)n * (n + 1) / 2;
    s.x = p; p += n; s.np = p; p += n; s.d = p; p += n; s.z = p; p += n; s.r = p; p += n; s.e = p; p += n;
    s.u = p; p += n + 1;
    s.cval = p; p += n + nc;
    s.cs_c = p; p += n; s.cs_s = p; p += n;
    s.sc = p; p += SD_COUNT;
    if ((p - reinterpret_cast<double*>(base)) & 1) ++p;
    int* q = reinterpret_cast<int*>(p);
    s.aset = q; q += n; s.aside = q; q += n; s.bstate = q; q += n;
    s.rstate = q; q += nc;
    s.cside = q; q += n + nc;
    s.si = q;
    return s;
}

OSOT_BIG_FN double clamp_inf(double v) { return v >= kInf ? kInf : (v <= -kInf ? -kInf : v); }
OSOT_BIG_FN int ridx(int i, int j) { return ((j * (j + 1)) >> 1) + i; }

#define OSOT_BIG_FOR(i, N) for (int i = tm.tid; i < (N); i += tm.nt)

// Team: { int tid, nt; void sync() const; }  -- every thread of the team calls solve() with the same arguments
template <class Team>
OSOT_BIG_FN void solve(const Team& tm, const Args& a, const Shared& s) {
    const int n = a.n, nc = a.nc;
    const bool t0 = tm.tid == 0;
    const bool has_box = a.l != nullptr;
    double* L = a.Lw;
    double* J = a.J;
    double* Rp = s.R;
    // ---- H + eps I -> L, state
    OSOT_BIG_FOR(e2, n * n) { const int i = e2 / n, j = e2 - i * n; L[e2] = a.H[e2] + ((i == j) ? a.eps : 0.0); }
    OSOT_BIG_FOR(i, n) { s.bstate[i] = 0; s.x[i] = 0.0; s.u[i] = 0.0; s.aset[i] = -1; s.aside[i] = 0; s.d[i] = -a.g[i]; }
    OSOT_BIG_FOR(r, nc) s.rstate[r] = 0;
    if (t0) { s.si[SI_IQ] = 0; s.si[SI_ME] = 0; s.si[SI_ST] = ST_SOLVED; s.si[SI_ITERS] = 0; s.si[SI_DONE] = 0; s.u[n] = 0.0; }
    tm.sync();
    auto finish = [&]() {              // (called by every thread, behind a barrier)
        const int st = s.si[SI_ST];
        OSOT_BIG_FOR(i, n) a.x[i] = (st == ST_SOLVED) ? s.x[i] : 0.0;
        if (t0) { *a.status = st; if (a.iters) *a.iters = s.si[SI_ITERS]; }
        tm.sync();
    };
    // ---- Cholesky of H + eps I, right-looking, in place (lower triangle of L)
    for (int k = 0; k < n; ++k) {
        if (t0) {
            const double p = L[k * n + k];
            if (!(p > 0.0) || !(p < INFINITY)) s.si[SI_ST] = ST_NOT_PD;
            else { const double sq = sqrt(p); L[k * n + k] = sq; s.sc[SD_PIV] = 1.0 / sq; s.d[k] *= 1.0 / sq; }      // (y_k: final)
        }
        tm.sync();
        if (s.si[SI_ST] != ST_SOLVED) { finish(); return; }
        const double ip = s.sc[SD_PIV];
        const int m = n - k - 1;
        const double yk = s.d[k];
        OSOT_BIG_FOR(i2, m) { const int i = k + 1 + i2; const double lik = L[i * n + k] * ip; L[i * n + k] = lik; s.d[i] -= lik * yk; }
        tm.sync();
        OSOT_BIG_FOR(e2, m * m) {
            const int ii = e2 / m, jj = e2 - ii * m;
            if (jj <= ii) { const int i = k + 1 + ii, j = k + 1 + jj; L[i * n + j] -= L[i * n + k] * L[j * n + k]; }
        }
        tm.sync();
    }
    // ---- J = L^-T: thread j computes column j of L^-1 by forward substitution = row j of J
    OSOT_BIG_FOR(j, n) {
        double* row = J + (size_t)j * n;
        for (int i = 0; i < j; ++i) row[i] = 0.0;
        row[j] = 1.0 / L[j * n + j];
        for (int i = j + 1; i < n; ++i) {
            double acc = 0.0;
            for (int k = j; k < i; ++k) acc += L[i * n + k] * row[k];
            row[i] = -acc / L[i * n + i];
        }
    }
    tm.sync();
    // ---- unconstrained minimiser: L y = -g rode along with the factorisation (s.d = y), x = L^-T y = J y.  (Forward SUBSTITUTION, not
    // J'g: on a rank-deficient H + eps I with g in range(H) it keeps the cancellation in the eps-pivots -- the explicit inverse put 1e-6 of
    // null-space noise into x where substitution leaves 5e-8: tests/stress_qp.py "wide", the known-answer structure of TestQPOases.cpp:274-340)
    OSOT_BIG_FOR(i, n) { double acc = 0.0; for (int j = i; j < n; ++j) acc += J[i * n + j] * s.d[j]; s.x[i] = acc; }
    tm.sync();

    // d = J'n for the constraint (code, side), |d|^2 and |d2|^2, z = J2 d2.  Ends behind a barrier.
    auto compute_d = [&](int code, int side) {
        const int iq = s.si[SI_IQ];
        if (code < n) {
            OSOT_BIG_FOR(j, n) s.d[j] = side * J[code * n + j];
        } else {
            const double* ar = a.A + (size_t)(code - n) * n;
            OSOT_BIG_FOR(i, n) s.np[i] = side * ar[i];
            tm.sync();
            OSOT_BIG_FOR(j, n) { double acc = 0.0; for (int i = 0; i < n; ++i) acc += J[i * n + j] * s.np[i]; s.d[j] = acc; }
        }
        tm.sync();
        if (t0) {
            double dd = 0.0, nd2 = 0.0;
            for (int j = 0; j < n; ++j) { const double v = s.d[j] * s.d[j]; dd += v; if (j >= iq) nd2 += v; }
            s.sc[SD_DD] = dd; s.sc[SD_ND2] = nd2;
        }
        OSOT_BIG_FOR(i, n) { double acc = 0.0; const double* row = J + (size_t)i * n; for (int j = iq; j < n; ++j) acc += row[j] * s.d[j]; s.z[i] = acc; }
        tm.sync();
    };
    // the constraint whose d, z are current enters the working set at position iq: one Householder reflection of J2.  Ends behind a barrier.
    auto add_constraint = [&](int code, int side) {
        const int iq = s.si[SI_IQ];
        if (t0) {
            const double nd2 = s.sc[SD_ND2], d0 = s.d[iq];
            const double nrm = sqrt(nd2), alpha = (d0 > 0.0) ? -nrm : nrm;
            s.sc[SD_ALPHA] = alpha; s.sc[SD_BETA] = 1.0 / (nd2 - alpha * d0);
            for (int i = 0; i < iq; ++i) Rp[ridx(i, iq)] = s.d[i];
            Rp[ridx(iq, iq)] = alpha;
            s.aset[iq] = code; s.aside[iq] = side;
            if (code < n) s.bstate[code] = (side > 0) ? 1 : 2;
            else if (s.rstate[code - n] != 3) s.rstate[code - n] = (side > 0) ? 1 : 2;
        }
        tm.sync();
        const double alpha = s.sc[SD_ALPHA], beta = s.sc[SD_BETA];
        const double v0 = s.d[iq] - alpha;
        OSOT_BIG_FOR(i, n) {
            double* row = J + (size_t)i * n;
            const double bw = beta * (s.z[i] - alpha * row[iq]);      // J2 v = J2 d2 - alpha J[:, iq]
            row[iq] -= bw * v0;
            for (int j = iq + 1; j < n; ++j) row[j] -= bw * s.d[j];
        }
        if (t0) s.si[SI_IQ] = iq + 1;
        tm.sync();
    };
    // the constraint at position l leaves the working set: column l out of R, Givens rotations restore the triangle (thread 0, column
    // by column through a scratch vector: the packed storage has no room for the sub-diagonal), the same rotations on the columns of J.
    auto drop_constraint = [&](int l) {
        const int iq = s.si[SI_IQ];
        if (t0) {
            const int code = s.aset[l];
            if (code < n) s.bstate[code] = 0; else s.rstate[code - n] = 0;
            double* e = s.e;
            for (int k = l; k < iq - 1; ++k) {
                for (int i = 0; i <= k + 1; ++i) e[i] = Rp[ridx(i, k + 1)];
                for (int j = l; j < k; ++j) {
                    const double c = s.cs_c[j], sn = s.cs_s[j], a0 = e[j], b0 = e[j + 1];
                    e[j] = c * a0 + sn * b0; e[j + 1] = -sn * a0 + c * b0;
                }
                const double a0 = e[k], b0 = e[k + 1];
                const double h = sqrt(a0 * a0 + b0 * b0);
                double c = 1.0, sn = 0.0;
                if (h > 0.0) { c = a0 / h; sn = b0 / h; }
                s.cs_c[k] = c; s.cs_s[k] = sn;
                e[k] = h;
                for (int i = 0; i <= k; ++i) Rp[ridx(i, k)] = e[i];
                s.aset[k] = s.aset[k + 1]; s.aside[k] = s.aside[k + 1]; s.u[k] = s.u[k + 1];
            }
            s.u[iq - 1] = s.u[iq]; s.u[iq] = 0.0;       // (the candidate's multiplier follows the set)
            s.aset[iq - 1] = -1;
            s.si[SI_IQ] = iq - 1;
        }
        tm.sync();
        OSOT_BIG_FOR(i, n) {
            double* row = J + (size_t)i * n;
            for (int j = l; j < iq - 1; ++j) {
                const double c = s.cs_c[j], sn = s.cs_s[j], a0 = row[j], b0 = row[j + 1];
                row[j] = c * a0 + sn * b0; row[j + 1] = -sn * a0 + c * b0;
            }
        }
        tm.sync();
    };

    // ---- equality rows first (lA == uA after the clamp): a signed step onto each, no multipliers, never dropped
    for (int r = 0; r < nc; ++r) {
        const double lo = clamp_inf(a.lA[r]), up = clamp_inf(a.uA[r]);
        if (!(lo == up)) continue;
        tm.sync();                                       // (the scalars of the previous row have been read by everybody)
        if (t0) { double acc = 0.0; const double* ar = a.A + (size_t)r * n; for (int i = 0; i < n; ++i) acc += ar[i] * s.x[i]; s.sc[SD_SIP] = acc - lo; s.rstate[r] = 3; }
        compute_d(n + r, +1);
        const double nd2 = s.sc[SD_ND2], dd = s.sc[SD_DD], sip = s.sc[SD_SIP];
        if (!(nd2 > kDep2 * dd)) {                       // a combination of the rows already in
            if (fabs(sip) > kEq * fmax(1.0, fabs(lo))) { tm.sync(); if (t0) s.si[SI_ST] = ST_INFEASIBLE; tm.sync(); finish(); return; }
            continue;                                    // redundant and consistent
        }
        const double t = -sip / nd2;
        OSOT_BIG_FOR(i, n) s.x[i] += t * s.z[i];
        add_constraint(n + r, +1);
        if (t0) s.si[SI_ITERS] += 1;
    }
    tm.sync();
    if (t0) s.si[SI_ME] = s.si[SI_IQ];
    tm.sync();
    const int me = s.si[SI_ME];

    // ---- the dual loop
    for (;;) {
        // most violated inactive constraint
        OSOT_BIG_FOR(k, n) {
            double v = 0.0; int sd = 0;
            if (has_box && s.bstate[k] == 0) {
                const double lo = clamp_inf(a.l[k]), up = clamp_inf(a.u[k]), xk = s.x[k];
                if (lo > -kInf) { const double vi = lo - xk; if (vi > kViol * fmax(1.0, fabs(lo)) && vi > v) { v = vi; sd = +1; } }
                if (up < kInf) { const double vi = xk - up; if (vi > kViol * fmax(1.0, fabs(up)) && vi > v) { v = vi; sd = -1; } }
            }
            s.cval[k] = v; s.cside[k] = sd;
        }
        OSOT_BIG_FOR(r, nc) {
            double v = 0.0; int sd = 0;
            if (s.rstate[r] == 0) {
                const double lo = clamp_inf(a.lA[r]), up = clamp_inf(a.uA[r]);
                if (lo > -kInf || up < kInf) {
                    const double* ar = a.A + (size_t)r * n;
                    double ax = 0.0;
                    for (int i = 0; i < n; ++i) ax += ar[i] * s.x[i];
                    if (lo > -kInf) { const double vi = lo - ax; if (vi > kViol * fmax(1.0, fabs(lo)) && vi > v) { v = vi; sd = +1; } }
                    if (up < kInf) { const double vi = ax - up; if (vi > kViol * fmax(1.0, fabs(up)) && vi > v) { v = vi; sd = -1; } }
                }
            }
            s.cval[n + r] = v; s.cside[n + r] = sd;
        }
        tm.sync();
        if (t0) {
            double best = 0.0; int ip = -1;
            for (int k = 0; k < n + nc; ++k) if (s.cval[k] > best) { best = s.cval[k]; ip = k; }
            if (ip < 0) s.si[SI_DONE] = 1;
            else {
                const int sd = s.cside[ip];
                s.si[SI_IP] = ip; s.si[SI_SIDE] = sd; s.sc[SD_SIP] = -best;
                s.sc[SD_BND] = (ip < n) ? ((sd > 0) ? clamp_inf(a.l[ip]) : clamp_inf(a.u[ip])) : ((sd > 0) ? clamp_inf(a.lA[ip - n]) : clamp_inf(a.uA[ip - n]));
                s.u[s.si[SI_IQ]] = 0.0;
            }
        }
        tm.sync();
        if (s.si[SI_DONE]) break;
        const int ip = s.si[SI_IP], side = s.si[SI_SIDE];
        // the step(s) that bring constraint ip in
        for (;;) {
            compute_d(ip, side);
            if (t0) {
                const int iq = s.si[SI_IQ];
                for (int j = iq - 1; j >= 0; --j) {              // r = R^-1 d1
                    double acc = s.d[j];
                    for (int k = j + 1; k < iq; ++k) acc -= Rp[ridx(j, k)] * s.r[k];
                    s.r[j] = acc / Rp[ridx(j, j)];
                }
                double rmax = 0.0;
                for (int k = me; k < iq; ++k) rmax = fmax(rmax, fabs(s.r[k]));
                double t1 = INFINITY; int l = -1;
                for (int k = me; k < iq; ++k)
                    if (s.r[k] > kRatio * rmax && s.r[k] > 0.0) { const double q = s.u[k] / s.r[k]; if (q < t1) { t1 = q; l = k; } }
                const double nd2 = s.sc[SD_ND2], dd = s.sc[SD_DD], sip = s.sc[SD_SIP];
                const bool has_dir = nd2 > kDep2 * dd;
                const double t2 = has_dir ? -sip / nd2 : INFINITY;
                int act; double t;
                if (!(t1 < INFINITY) && !(t2 < INFINITY)) { act = 0; t = 0.0; }          // no step at all: infeasible
                else if (t2 <= t1) { act = 2; t = t2; }                                    // full step: ip enters
                else if (!has_dir) { act = 1; t = t1; }                                    // dual step only: l leaves
                else { act = 3; t = t1; }                                                  // partial step: l leaves
                if (act != 0) {
                    for (int k = 0; k < iq; ++k) s.u[k] -= t * s.r[k];
                    s.u[iq] += t;
                }
                if (++s.si[SI_ITERS] > a.max_iter) act = 4;
                s.si[SI_ACT] = act; s.si[SI_L] = l; s.sc[SD_T] = t;
            }
            tm.sync();
            const int act = s.si[SI_ACT];
            const double t = s.sc[SD_T];
            if (act == 0 || act == 4) { if (t0) s.si[SI_ST] = (act == 0) ? ST_INFEASIBLE : ST_MAX_ITER; tm.sync(); finish(); return; }
            if (act >= 2) { OSOT_BIG_FOR(i, n) s.x[i] += t * s.z[i]; }
            tm.sync();
            if (act == 2) { add_constraint(ip, side); break; }
            drop_constraint(s.si[SI_L]);
            if (act == 3) {                                      // the candidate's slack at the new x
                if (t0) {
                    double ax;
                    if (ip < n) ax = s.x[ip];
                    else { ax = 0.0; const double* ar = a.A + (size_t)(ip - n) * n; for (int i = 0; i < n; ++i) ax += ar[i] * s.x[i]; }
                    s.sc[SD_SIP] = side * (ax - s.sc[SD_BND]);
                }
                tm.sync();
            }
        }
    }
    finish();
}

}  // namespace big

#if defined(__HIPCC__) && !defined(OSOT_BIG_HOST) && !defined(OSOT_EMULATION)
// ---- the device side: one 256-thread workgroup per QP; a launch of G workgroups walks the batch with stride G, each workgroup on its
// own slice of the workspace (2 n^2 doubles)
struct DevQPBig {
    int B, n, nc, max_iter;
    double eps_abs;
    const double *H, *g, *A, *lA, *uA, *l, *u;
    double* x;
    int* status;
    int* iterations;
    double* work;                     // [gridDim.x][2][n][n]
};
struct BigTeamDev {
    int tid, nt;
    __device__ void sync() const { __syncthreads(); }
};
__global__ void __launch_bounds__(256) osot_qp_big_kernel(const DevQPBig Q) {
    extern __shared__ __attribute__((aligned(16))) char osot_big_smem[];
    const int n = Q.n, nc = Q.nc;
    const big::Shared sh = big::carve(osot_big_smem, n, nc);
    const BigTeamDev tm{(int)threadIdx.x, (int)blockDim.x};
    double* slot = Q.work + (size_t)blockIdx.x * 2 * (size_t)n * n;
    for (long long inst = blockIdx.x; inst < Q.B; inst += gridDim.x) {
        big::Args a;
        a.n = n; a.nc = nc; a.max_iter = Q.max_iter; a.eps = Q.eps_abs;
        a.H = Q.H + inst * (long long)n * n; a.g = Q.g + inst * n;
        a.A = nc ? Q.A + inst * (long long)nc * n : nullptr;
        a.lA = nc ? Q.lA + inst * nc : nullptr; a.uA = nc ? Q.uA + inst * nc : nullptr;
        a.l = Q.l ? Q.l + inst * n : nullptr; a.u = Q.u ? Q.u + inst * n : nullptr;
        a.x = Q.x + inst * n; a.status = Q.status + inst; a.iters = Q.iterations ? Q.iterations + inst : nullptr;
        a.Lw = slot; a.J = slot + (size_t)n * n;
        big::solve(tm, a, sh);
        __syncthreads();
    }
}
#endif
}  // namespace osot
