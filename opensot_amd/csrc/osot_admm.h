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
// 4000 iterations at most.  No Ruiz equilibration (the dense n <= 64 systems here are factorised exactly each time rho
// moves, which is what the scaling buys osqp's sparse LDL').
//
// One wavefront per QP, lane = variable (n <= 64).  LDS: the explicit inverse of the n x n system matrix and the rows of A
// (row stride n + 1... odd or not, the walks below touch one column or one row per step: conflict-free), the iterates.
#pragma once
#include <osot_team.h>
#include <osot_mi355x.h>

namespace osot {

struct DevAdmm {
    int B, n, nc, max_iter, check_every;
    double eps_reg, eps_abs, eps_rel, rho0, sigma, alpha;
    const double* H; const double* g; const double* A; const double* lA; const double* uA; const double* l; const double* u;
    double* x; int* status; int* iterations;
};

// lane-c value summed / maxed over the 64 lanes
__device__ __forceinline__ double wsum64(double v) { return colsum<64>(v); }
__device__ __forceinline__ double wmax64(double v) { return colmax<64>(v); }

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
    const double q = valid ? Q.g[inst * n + c] : 0.0;
    wave_sync();
    double rho = Q.rho0;
    int status = 2, it = 0;        // OSOT_STATUS_MAX_ITER unless something better happens
    bool refactor = true;
    const double qinf = wmax64(fabs(q));
    while (it < Q.max_iter) {
        if (refactor) {
            // M = P + sigma I + A' R_A A + R_I  (lane c = column c), then in-place Gauss-Jordan inversion (M is SPD)
            for (int i = 0; i < n; ++i) {
                double v = valid ? Hg[i * n + c] + ((i == c) ? Q.eps_reg + Q.sigma + (has_box ? rr[nc + c] : 0.0) : 0.0) : 0.0;
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
        if (has_box && valid) rhs += rr[nc + c] * zs[nc + c] - ys[nc + c];
        if (valid) rh[c] = rhs;
        wave_sync();
        double xtc = 0.0;
        if (valid) for (int j = 0; j < n; ++j) xtc = fma(Mi[c * S + j], rh[j], xtc);
        if (valid) xt[c] = xtc;
        wave_sync();
        // zt = Abar xt  (lane = row for the A part, lane = variable for the box part)
        for (int r = c; r < nc; r += 64) { double a = 0.0; for (int j = 0; j < n; ++j) a = fma(Al[r * S + j], xt[j], a); zt[r] = a; }
        if (has_box && valid) zt[nc + c] = xtc;
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
            else ax = xs[r - nc];
            axn = fmax(axn, fabs(ax)); zn_ = fmax(zn_, fabs(zs[r])); rp = fmax(rp, fabs(ax - zs[r]));
        }
        axn = wmax64(axn); zn_ = wmax64(zn_); rp = wmax64(rp);
        double px = 0.0, aty = 0.0;
        if (valid) {
            for (int j = 0; j < n; ++j) px = fma(Hg[c * n + j] + ((j == c) ? Q.eps_reg : 0.0), xs[j], px);
            for (int r = 0; r < nc; ++r) aty = fma(Al[r * S + c], ys[r], aty);
            if (has_box) aty += ys[nc + c];
        }
        const double rd = wmax64(fabs(px + q + aty));
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
                const double yr = ys[r];
                yn = fmax(yn, fabs(yr));
                if (yr > 0.0 && up[r] < 1.0e20) sup += up[r] * yr;
                else if (yr < 0.0 && lo[r] > -1.0e20) sup += lo[r] * yr;
            }
            yn = wmax64(yn); sup = wsum64(sup);
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
    if (valid) Q.x[inst * n + c] = (status == 0) ? xs[c] : 0.0;
    if (c == 0) { Q.status[inst] = status; if (Q.iterations) Q.iterations[inst] = it; }
}

inline size_t admm_lds_bytes(int n, int nc, bool has_box) {
    const int m = nc + (has_box ? n : 0);
    return sizeof(double) * ((size_t)(n + nc) * (n + 1) + 3 * (size_t)n + 6 * (size_t)m + 8);
}

}  // namespace osot
