// opensot_amd/csrc/osot_admm.h -- second back-end: an OSQP-CONVENTION ADMM solver for B generic QPs.
//
// Problem in BackEnd convention (include/OpenSoT/solvers/BackEnd.h:125-150), posed the way the reference's OSQP back-end
// poses it (src/solvers/OSQPBackEnd.cpp:25-49, 120-143, 198-226):  P = H + eps I with eps = 2.22e-13 * factor (:8, 29),
// ONE constraint matrix  Abar = [A; I]  with the rows' and the box's bounds piled  l_bar = [lA; l], u_bar = [uA; u]
// (:62-64: missing sides are -1 / 1 placeholders there, +-1e20 "absent" here), OSQP's default settings except
// eps_abs = eps_rel = 1e-5 (:38-39); "solved" and "solved inaccurate" both count as success (:213-216).
//
// Algorithm: the ADMM iteration of OSQP (Stellato et al., "OSQP: an operator splitting solver for quadratic programs", Math.
// Prog. Comp. 12 (2020), Algorithm 1; osqp is NOT vendored in the reference -- docs suggest release 0.6.2 -- so this is a
// restatement of the published algorithm, PARITY UNPINNED against osqp itself):
//     solve (P + sigma I + Abar' R Abar) xt = sigma x - q + Abar'(R z - y)          R = diag(rho_i), rho_eq = 1e3 rho
//     zt = Abar xt;  x <- a xt + (1 - a) x;  z <- clip(a zt + (1 - a) z + y / rho);  y <- y + rho (a zt + (1 - a) z_old - z)
// with sigma = 1e-6, alpha = 1.6, rho = 0.1, the residual test every 25 iterations on UNSCALED residuals, adaptive rho
// (ratio of the normalised residuals; refactorisation when it moves by more than 5x), primal infeasibility certificate,
// 4000 iterations at most.
// Ruiz equilibration as osqp does it before the first iteration (scaling = 10 passes; Algorithm 2 of the paper): diagonal
// D (variables), E (rows), cost factor c with  Pbar = c D P D, qbar = c D q, Abar <- E Abar D, bounds <- E bounds; the
// iteration runs on the scaled problem, the residual test on the UNSCALED residuals (scaled_termination = 0), x = D xbar,
// y = E ybar / c.  Pbar is never stored: the system matrix and the dual residual apply c d_i d_j to H as they read it; A is
// scaled in place in LDS; the identity block of Abar keeps one entry per row, s_j = e_(nc+j) d_j.
// Warm start (OSQPBackEnd keeps its workspace from one control cycle to the next, OSQPBackEnd.cpp:120-143, 268-287, and osqp
// starts from the previous x, y with the rho it ended on): an optional per-instance state (x, y, rho), read before and
// written after the solve; z starts at clip(Abar x).
//
// One wavefront per QP, lane = variable (n <= 64).  LDS: the explicit inverse of the n x n system matrix and the rows of A
// (row stride n + 1... odd or not, the walks below touch one column or one row per step: conflict-free), the iterates.
#pragma once
#include <cstring>
#include <osot_team.h>
#include <osot_mi355x.h>

namespace osot {

struct DevAdmm {
    int B, n, nc, max_iter, check_every;
    double eps_reg, eps_abs, eps_rel, rho0, sigma, alpha;
    const double* H; const double* g; const double* A; const double* lA; const double* uA; const double* l; const double* u;
    double* x; int* status; int* iterations;
    int scaling;            // Ruiz passes (osqp default 10; 0 = none)
    double* warm_x;         // [B][n]   in/out, UNSCALED; null = cold start every call
    double* warm_y;         // [B][nc + n] (rows of A, then the box rows; the box part is absent without l/u)
    double* warm_rho;       // [B]      rho the previous solve ended on; <= 0 = no state yet (cold start of that instance)
};

// lane-c value summed / maxed over the 64 lanes
// (uniform_d: the results are wave-uniform and the iteration's exits depend on them -- said so, the loop's control flow and its
//  counters stay on the scalar unit: osot_team.h)
__device__ __forceinline__ double wsum64(double v) { return uniform_d(colsum<64>(v)); }
__device__ __forceinline__ double wmax64(double v) { return uniform_d(colmax<64>(v)); }

__global__ void __launch_bounds__(64) osot_admm_kernel(const DevAdmm Q) {
    OSOT_DYNAMIC_LDS(admm_smem);
    const long long inst = blockIdx.x;
    const int c = threadIdx.x;
    if (inst >= Q.B) return;
    const int n = Q.n, nc = Q.nc, S = n + 1;
    const bool has_box = Q.l != nullptr;
    const int m = nc + (has_box ? n : 0);
    double* Mi = reinterpret_cast<double*>(admm_smem);   // [n][S] system matrix -> its inverse
    double* Al = Mi + n * S;                             // [nc][S] rows of A
    double* xs = Al + nc * S;                            // x, xt, rhs: n each
    double* xt = xs + n;
    double* rh = xt + n;
    double* zs = rh + n;                                 // z, y, zt, lo, up, rho: m each
    double* ys = zs + m;
    double* zt = ys + m;
    double* lo = zt + m;
    double* up = lo + m;
    double* rr = up + m;
    double* es = rr + m;                                 // E (rows of A, then the box rows)
    const bool valid = c < n;
    // ---- data -> LDS
    const double* Hg = Q.H + inst * (long long)n * n;
    const double* Ag = Q.A ? Q.A + inst * (long long)nc * n : nullptr;
    for (int r = 0; r < nc; ++r) if (valid) Al[r * S + c] = Ag[r * n + c];
    for (int r = c; r < m; r += 64) {
        double a, b;
        if (r < nc) { a = Q.lA[inst * nc + r]; b = Q.uA[inst * nc + r]; }
        else { a = Q.l[inst * n + r - nc]; b = Q.u[inst * n + r - nc]; }
        a = a < -1.0e20 ? -1.0e20 : (a > 1.0e20 ? 1.0e20 : a);
        b = b < -1.0e20 ? -1.0e20 : (b > 1.0e20 ? 1.0e20 : b);
        lo[r] = a; up[r] = b;
        zs[r] = 0.0; ys[r] = 0.0; zt[r] = 0.0;
        rr[r] = (a == b) ? 1.0e3 * Q.rho0 : ((a <= -1.0e20 && b >= 1.0e20) ? 1.0e-6 : Q.rho0);   // osqp: rho_eq, RHO_MIN for free rows
    }
    if (valid) { xs[c] = 0.0; xt[c] = 0.0; }
    const double q0 = valid ? Q.g[inst * n + c] : 0.0;   // unscaled q
    for (int r = c; r < m; r += 64) es[r] = 1.0;
    wave_sync();
    // ---- Ruiz equilibration (osqp scaling.c: column norms of the KKT matrix [P A'; A 0] in the infinity norm, entries
    //      below 1e-4 count as 1, factors kept within [1e-4, 1e4]; then the cost factor from the mean column norm of P and |q|)
    double dsc = 1.0, csc = 1.0;   // lane c: d_c;  c
    auto limit = [](double v) { v = (v < 1.0e-4) ? 1.0 : v; return v > 1.0e4 ? 1.0e4 : v; };
    for (int pass = 0; pass < Q.scaling; ++pass) {
        if (valid) xt[c] = dsc;                           // d, for the other lanes' column norms of P
        wave_sync();
        double pn = 0.0, an = 0.0;
        if (valid) {
            for (int i = 0; i < n; ++i) pn = fmax(pn, fabs(Hg[i * n + c] + ((i == c) ? Q.eps_reg : 0.0)) * xt[i]);
            pn *= csc * dsc;
            for (int r = 0; r < nc; ++r) an = fmax(an, fabs(Al[r * S + c]));
            if (has_box) an = fmax(an, es[nc + c] * dsc);
        }
        const double dt = valid ? 1.0 / sqrt(limit(fmax(pn, an))) : 1.0;
        // row norms of Abar (lane = row), then E_temp
        for (int r = c; r < m; r += 64) {
            double rn = 0.0;
            if (r < nc) { for (int j = 0; j < n; ++j) rn = fmax(rn, fabs(Al[r * S + j])); }
            else rn = es[r] * xt[r - nc];
            zt[r] = 1.0 / sqrt(limit(rn));
        }
        wave_sync();
        for (int r = 0; r < nc; ++r) if (valid) Al[r * S + c] *= zt[r] * dt;
        wave_sync();
        for (int r = c; r < m; r += 64) es[r] *= zt[r];
        dsc *= dt;
        wave_sync();
        // cost scaling: c_temp = 1 / max(mean_j |Pbar[:, j]|_inf, |qbar|_inf)
        if (valid) xt[c] = dsc;
        wave_sync();
        double pc = 0.0;
        if (valid) { for (int i = 0; i < n; ++i) pc = fmax(pc, fabs(Hg[i * n + c] + ((i == c) ? Q.eps_reg : 0.0)) * xt[i]); pc *= csc * dsc; }
        const double pmean = wsum64(valid ? pc : 0.0) / (double)n;
        const double qn = wmax64(fabs(csc * dsc * q0));
        csc *= 1.0 / limit(fmax(pmean, qn));
        wave_sync();
    }
    if (valid) { xt[c] = dsc; }                            // d stays in LDS for the products with P below
    double* dv = rr + 2 * m;                               // [n] copy of d that nothing else overwrites
    if (valid) dv[c] = dsc;
    wave_sync();
    for (int r = c; r < m; r += 64) { lo[r] = (lo[r] <= -1.0e20) ? -1.0e20 : lo[r] * es[r]; up[r] = (up[r] >= 1.0e20) ? 1.0e20 : up[r] * es[r]; zt[r] = 0.0; }
    const double sbx = has_box && valid ? es[nc + c] * dsc : 0.0;   // the box row's single entry of Abar
    const double q = csc * dsc * q0;
    // ---- warm start
    double rho = Q.rho0;
    const bool warm = Q.warm_x && Q.warm_rho && Q.warm_rho[inst] > 0.0;
    if (warm) {
        rho = Q.warm_rho[inst];
        if (valid) xs[c] = Q.warm_x[inst * n + c] / dsc;
        wave_sync();
        for (int r = c; r < m; r += 64) {
            double ax;
            if (r < nc) { ax = 0.0; for (int j = 0; j < n; ++j) ax = fma(Al[r * S + j], xs[j], ax); }
            else ax = es[r] * dv[r - nc] * xs[r - nc];
            zs[r] = ax < lo[r] ? lo[r] : (ax > up[r] ? up[r] : ax);
            ys[r] = Q.warm_y ? Q.warm_y[inst * m + r] * csc / es[r] : 0.0;
            const bool eq = lo[r] == up[r], fre = (lo[r] <= -1.0e20 && up[r] >= 1.0e20);
            rr[r] = eq ? 1.0e3 * rho : (fre ? 1.0e-6 : rho);
        }
    }
    wave_sync();
    int status = 2, it = 0;        // OSOT_STATUS_MAX_ITER unless something better happens
    bool refactor = true;
    const double qinf = wmax64(fabs(q0));
    while (it < Q.max_iter) {
        if (refactor) {
            // M = P + sigma I + A' R_A A + R_I  (lane c = column c), then in-place Gauss-Jordan inversion (M is SPD)
            for (int i = 0; i < n; ++i) {
                double v = valid ? csc * dv[i] * dsc * (Hg[i * n + c] + ((i == c) ? Q.eps_reg : 0.0)) + ((i == c) ? Q.sigma + (has_box ? rr[nc + c] * sbx * sbx : 0.0) : 0.0) : 0.0;
                for (int r = 0; r < nc; ++r) v = fma(rr[r] * Al[r * S + i], valid ? Al[r * S + c] : 0.0, v);
                if (valid) Mi[i * S + c] = v;
            }
            wave_sync();
            for (int k = 0; k < n; ++k) {
                const double ip = 1.0 / Mi[k * S + k];
                const double rowk = valid ? Mi[k * S + c] * ip : 0.0;    // scaled pivot row at my column
                wave_sync();                                              // pivot row read by everyone before it is rewritten
                if (valid && c != k) {
                    for (int i = 0; i < n; ++i)
                        if (i != k) Mi[i * S + c] = fma(-Mi[i * S + k], rowk, Mi[i * S + c]);
                    Mi[k * S + c] = rowk;
                }
                wave_sync();                                              // column k read by everyone before lane k rewrites it
                if (c == k) for (int i = 0; i < n; ++i) Mi[i * S + k] = (i == k) ? ip : -Mi[i * S + k] * ip;
                wave_sync();
            }
            refactor = false;
        }
        // ---- one ADMM iteration
        // rhs = sigma x - q + A'(R z - y)_A + (R z - y)_I
        double rhs = valid ? Q.sigma * xs[c] - q : 0.0;
        for (int r = 0; r < nc; ++r) rhs = fma(valid ? Al[r * S + c] : 0.0, rr[r] * zs[r] - ys[r], rhs);
        if (has_box && valid) rhs += sbx * (rr[nc + c] * zs[nc + c] - ys[nc + c]);
        if (valid) rh[c] = rhs;
        wave_sync();
        double xtc = 0.0;
        if (valid) for (int j = 0; j < n; ++j) xtc = fma(Mi[c * S + j], rh[j], xtc);
        if (valid) xt[c] = xtc;
        wave_sync();
        // zt = Abar xt  (lane = row for the A part, lane = variable for the box part)
        for (int r = c; r < nc; r += 64) { double a = 0.0; for (int j = 0; j < n; ++j) a = fma(Al[r * S + j], xt[j], a); zt[r] = a; }
        if (has_box && valid) zt[nc + c] = sbx * xtc;
        const double xo = valid ? xs[c] : 0.0;
        const double xn = Q.alpha * xtc + (1.0 - Q.alpha) * xo;
        if (valid) xs[c] = xn;
        wave_sync();
        for (int r = c; r < m; r += 64) {
            const double zr = Q.alpha * zt[r] + (1.0 - Q.alpha) * zs[r];
            double zn = zr + ys[r] / rr[r];
            zn = zn < lo[r] ? lo[r] : (zn > up[r] ? up[r] : zn);
            ys[r] += rr[r] * (zr - zn);
            zs[r] = zn;
        }
        wave_sync();
        ++it;
        if (it % Q.check_every != 0 && it < Q.max_iter) continue;
        // ---- residuals (unscaled): r_prim = |Abar x - z|, r_dual = |P x + q + Abar'y|
        double axn = 0.0, zn_ = 0.0, rp = 0.0;
        for (int r = c; r < m; r += 64) {
            double ax;
            if (r < nc) { ax = 0.0; for (int j = 0; j < n; ++j) ax = fma(Al[r * S + j], xs[j], ax); }
            else ax = es[r] * dv[r - nc] * xs[r - nc];
            const double ei = 1.0 / es[r];                 // E^-1: the residual and its norms in the caller's units
            axn = fmax(axn, fabs(ax) * ei); zn_ = fmax(zn_, fabs(zs[r]) * ei); rp = fmax(rp, fabs(ax - zs[r]) * ei);
        }
        axn = wmax64(axn); zn_ = wmax64(zn_); rp = wmax64(rp);
        double px = 0.0, aty = 0.0;
        if (valid) {
            // (P x)_c, (Abar' y)_c of the scaled problem, brought back by D^-1 / c:  P x = H (D xbar),  A'y = (Abar'ybar) / (c d_c)
            for (int j = 0; j < n; ++j) px = fma(Hg[c * n + j] + ((j == c) ? Q.eps_reg : 0.0), dv[j] * xs[j], px);
            for (int r = 0; r < nc; ++r) aty = fma(Al[r * S + c], ys[r], aty);
            if (has_box) aty += sbx * ys[nc + c];
            aty /= csc * dsc;
        }
        const double rd = wmax64(fabs(px + q0 + aty));
        const double pxn = wmax64(fabs(px)), atyn = wmax64(fabs(aty));
        const double eps_p = Q.eps_abs + Q.eps_rel * fmax(axn, zn_);
        const double eps_d = Q.eps_abs + Q.eps_rel * fmax(fmax(pxn, atyn), qinf);
        if (rp <= eps_p && rd <= eps_d) { status = 0; break; }
        if (it >= Q.max_iter) {      // "solved inaccurate" (10x the tolerances) is a success for OSQPBackEnd::solve (:213-216)
            if (rp <= 10.0 * eps_p && rd <= 10.0 * eps_d) status = 0;
            break;
        }
        // primal infeasibility certificate on y itself once it has grown: |Abar'y| <= eps |y| and u'y+ + l'y- < 0
        {
            double yn = 0.0, sup = 0.0;
            for (int r = c; r < m; r += 64) {
                const double yr = ys[r];                   // (unscaled: y = E ybar / c, u'y = ubar'ybar / c)
                yn = fmax(yn, fabs(yr) * es[r]);
                if (yr > 0.0 && up[r] < 1.0e20) sup += up[r] * yr;
                else if (yr < 0.0 && lo[r] > -1.0e20) sup += lo[r] * yr;
            }
            yn = wmax64(yn) / csc; sup = wsum64(sup) / csc;
            if (yn > 1.0e6 && atyn <= 1.0e-4 * yn && sup < -1.0e-4 * yn) { status = 1; break; }
        }
        // adaptive rho (osqp: rho <- rho sqrt(normalised primal / normalised dual residual), refactor beyond 5x)
        {
            const double np_ = rp / fmax(fmax(axn, zn_), 1.0e-10), nd_ = rd / fmax(fmax(fmax(pxn, atyn), qinf), 1.0e-10);
            double rn = rho * sqrt(np_ / fmax(nd_, 1.0e-30));
            rn = rn < 1.0e-6 ? 1.0e-6 : (rn > 1.0e6 ? 1.0e6 : rn);
            if (it % (2 * Q.check_every) == 0 && (rn > 5.0 * rho || rn < 0.2 * rho)) {
                for (int r = c; r < m; r += 64) {
                    const bool eq = lo[r] == up[r], fre = (lo[r] <= -1.0e20 && up[r] >= 1.0e20);
                    rr[r] = eq ? 1.0e3 * rn : (fre ? 1.0e-6 : rn);
                }
                rho = rn;
                refactor = true;
                wave_sync();
            }
        }
    }
    if (valid) Q.x[inst * n + c] = (status == 0) ? dsc * xs[c] : 0.0;
    if (Q.warm_x && Q.warm_rho) {   // the state for the next solve of this instance (a failed solve leaves it cold)
        if (valid) Q.warm_x[inst * n + c] = dsc * xs[c];
        if (Q.warm_y) for (int r = c; r < m; r += 64) Q.warm_y[inst * m + r] = ys[r] * es[r] / csc;
        if (c == 0) Q.warm_rho[inst] = (status == 0) ? rho : 0.0;
    }
    if (c == 0) { Q.status[inst] = status; if (Q.iterations) Q.iterations[inst] = it; }
}

// DevAdmm from the C-ABI's arguments (osot_mi355x.hip and the emulator driver share it): defaults of OSQPBackEnd.cpp:36-39 + osqp
inline DevAdmm admm_args(int B, int n, int nc, const double* H, const double* g, const double* A, const double* lA,
                         const double* uA, const double* l, const double* u, double eps_reg, const osot_admm_options* opt,
                         double* warm_x, double* warm_y, double* warm_rho, double* x, int* status, int* iterations) {
    DevAdmm Q;
    std::memset(&Q, 0, sizeof(Q));
    Q.B = B; Q.n = n; Q.nc = nc;
    Q.max_iter = (opt && opt->max_iter > 0) ? opt->max_iter : 4000;
    Q.check_every = (opt && opt->check_every > 0) ? opt->check_every : 25;
    Q.eps_reg = eps_reg;
    Q.eps_abs = (opt && opt->eps_abs > 0.0) ? opt->eps_abs : 1.0e-5;
    Q.eps_rel = (opt && opt->eps_rel > 0.0) ? opt->eps_rel : 1.0e-5;
    Q.rho0 = (opt && opt->rho > 0.0) ? opt->rho : 0.1;
    Q.sigma = (opt && opt->sigma > 0.0) ? opt->sigma : 1.0e-6;
    Q.alpha = (opt && opt->alpha > 0.0) ? opt->alpha : 1.6;
    Q.scaling = !opt || opt->scaling == 0 ? 10 : (opt->scaling < 0 ? 0 : opt->scaling);
    Q.H = H; Q.g = g; Q.A = A; Q.lA = lA; Q.uA = uA; Q.l = l; Q.u = u; Q.x = x; Q.status = status; Q.iterations = iterations;
    Q.warm_x = warm_x; Q.warm_y = warm_y; Q.warm_rho = warm_rho;
    return Q;
}

inline size_t admm_lds_bytes(int n, int nc, bool has_box) {
    const int m = nc + (has_box ? n : 0);
    return sizeof(double) * ((size_t)(n + nc) * (n + 1) + 4 * (size_t)n + 8 * (size_t)m + 8);   // + E (m), a spare m, d (n)
}

}  // namespace osot
