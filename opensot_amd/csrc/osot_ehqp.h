// opensot_amd/csrc/osot_ehqp.h -- the EQUALITY-ONLY front-end of the reference, OpenSoT::solvers::eHQP (src/solvers/eHQP.cpp),
// for B instances, all levels of an instance in ONE launch by one wavefront (no QP: the method is a chain of damped
// pseudo-inverses and projectors).
//
//   eHQP::solve          eHQP.cpp:64-95     x = 0, P_0 = I;  per level:  L = chol(W),  JP = L' A P,  thin SVD of JP,
//                                           x += JP^+ (L' b - L' A x),  P <- P - V V'
//   eHQP::getDampedPinv  eHQP.cpp:124-146   Sigma^+ over the first rank() values (Eigen: sigma_j >= sigma_min * sigma_max),
//                                           1 / sigma_j, or sigma_j / (sigma_j^2 + lambda^2) with lambda = min(sigma) when that
//                                           minimum is below sigma_min
//   (constraints, bounds and the linear term c of the stack are not used: eHQP.cpp:45-47, eHQP.h:35)
//
// Everything happens in the n-dimensional space:  JP' JP = P A'W A P = P H P  with H = A'W A the SAME matrix the iHQP
// cascade builds, and  JP^+ (L'b - L'A x) = V D V' P A'W (b - A x) = V D V' (g' - H x)  with g' = A'W b and
// D = diag(1 / sigma^2) (or 1 / (sigma^2 + lambda^2)) -- so neither the Cholesky factor of W nor the left singular vectors
// are ever formed: a dense weight enters through W A and W b (the update kernel's outputs), a diagonal one through w.
// The symmetric eigenproblem of K = P H P (n x n) is solved by the nHQP front-end's routine (sym_eig32: Householder
// tridiagonalisation + implicit QL): eigenvalues = sigma^2, eigenvectors = V.
// Resolution: squaring costs the singular values below ~1e-8 sigma_max; they are treated as zero (floor 1e-7 sigma_max,
// kEhqpFloor), which is what the reference's threshold makes of the exactly-zero ones (a projected Jacobian has at most
// rank(P) non-zero singular values: the trailing ones are zeroed by COUNT as well, rem below) -- a level that is nearly
// singular inside its projected space (1e-12 .. 1e-7 sigma_max) is where the two differ, and where the reference's own
// answer is an amplification by 1 / sigma.
// One wavefront per instance, lane = c + 32 h; n <= 32.
#pragma once
#include <cstring>
#include "osot_host_plan.h"
#include "osot_nhqp.h"

namespace osot {

constexpr double kEhqpFloor = 1.0e-7;

struct DevEhqp {
    int B, n, L;
    unsigned active_mask;
    int m[OSOT_KMAX_LEVELS], ma[OSOT_KMAX_LEVELS];
    const double* A[OSOT_KMAX_LEVELS];    // [B][ma][n] stored rows
    const double* b[OSOT_KMAX_LEVELS];    // [B][m]
    const double* w[OSOT_KMAX_LEVELS];    // [B][m] diagonal of W (null: ones)
    const double* WA[OSOT_KMAX_LEVELS];   // [B][ma][n] W A of the stored rows (levels with a non-diagonal weight), else null
    const double* Wb[OSOT_KMAX_LEVELS];   // [B][m]
    double sigma_min;
    double* dq;          // [B][n]
    int* status;         // [B]
    int* iterations;     // [B] or null (always 0: there is no active set)
    double* x_levels;    // [B][L][n] or null
};

__global__ void __launch_bounds__(64) osot_ehqp_kernel(const DevEhqp Q) {
    OSOT_STATIC_LDS(double, P, 32 * kNS);
    OSOT_STATIC_LDS(double, K, 32 * kNS);
    OSOT_STATIC_LDS(double, E, 32 * kNS);
    OSOT_STATIC_LDS(double, Vv, 4 * 32);
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    const long long inst = blockIdx.x;
    const int n = Q.n;
    const bool valid = c < n;
    for (int e = lane; e < 32 * kNS; e += 64) P[e] = 0.0;
    wave_sync();
    if (h == 0 && valid) P[c * kNS + c] = 1.0;
    wave_sync();
    double x = 0.0;
    int rem = n;     // dimension of range(P): a projected Jacobian cannot have more non-zero singular values
    for (int k = 0; k < Q.L; ++k) {
        if (!((Q.active_mask >> k) & 1u)) {
            if (Q.x_levels && valid && h == 0) Q.x_levels[(inst * Q.L + k) * n + c] = x;
            continue;
        }
        const int m = Q.m[k], ma = Q.ma[k];
        const double* Ak = Q.A[k] ? Q.A[k] + inst * ma * n : nullptr;
        const double* bk = Q.b[k] + inst * m;
        const double* wk = Q.w[k] ? Q.w[k] + inst * m : nullptr;
        const bool dense = Q.WA[k] != nullptr;
        const double* WAk = dense ? Q.WA[k] + inst * ma * n : nullptr;
        const double* Wbk = dense ? Q.Wb[k] + inst * m : nullptr;
        // ---- H = A'W A (rows i = 2 t + h, column c, in registers) and g' = A'W b -------------------------------------
        double hacc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) hacc[t] = 0.0;
        double gp = 0.0;
        for (int r0 = 0; r0 < ma; r0 += 4) {
            double a[4], la[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + u;
                const bool in = r < ma;
                a[u] = (in && valid) ? Ak[r * n + c] : 0.0;
                if (dense) { la[u] = (in && valid) ? WAk[r * n + c] : 0.0; gp = fma(a[u], in ? Wbk[r] : 0.0, gp); }
                else { la[u] = (in ? (wk ? wk[r] : 1.0) : 0.0) * a[u]; gp = fma(la[u], in ? bk[r] : 0.0, gp); }
            }
            wave_sync();                    // the previous group's broadcasts are done
            if (h == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) Vv[u * 32 + c] = la[u];
            }
            wave_sync();
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < 16; ++t) hacc[t] = fma(Vv[u * 32 + 2 * t + h], a[u], hacc[t]);
        }
        if (m > ma && c < m - ma) {         // Postural block appended to the level: A = [I 0] (Postural.cpp:37), diagonal weights
            const double wi = wk ? wk[ma + c] : 1.0;
            gp = fma(wi, bk[ma + c], gp);
#pragma unroll
            for (int t = 0; t < 16; ++t) if (2 * t + h == c) hacc[t] += wi;
        }
        wave_sync();
        // ---- u = g' - H x  (H symmetric: (H x)_c = sum_i H[i][c] x_i over this half's rows, then the other half's) ----
        if (h == 0) Vv[c] = valid ? x : 0.0;
        wave_sync();
        double hx = 0.0;
#pragma unroll
        for (int t = 0; t < 16; ++t) hx = fma(hacc[t], Vv[2 * t + h], hx);
        hx = halfsum<32>(hx);
        const double uvec = valid ? gp - hx : 0.0;
        // ---- K = P H P ------------------------------------------------------------------------------------------------
#pragma unroll
        for (int t = 0; t < 16; ++t) K[(2 * t + h) * kNS + c] = hacc[t];
        wave_sync();
        double acc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = 0.0;
        for (int j = 0; j < n; ++j) {                    // (H P)[i][c] = sum_j H[i][j] P[j][c]
            const double pj = P[j * kNS + c];
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = fma(K[(2 * t + h) * kNS + j], pj, acc[t]);
        }
        wave_sync();
#pragma unroll
        for (int t = 0; t < 16; ++t) E[(2 * t + h) * kNS + c] = acc[t];
        wave_sync();
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = 0.0;
        for (int j = 0; j < n; ++j) {                    // (P (H P))[i][c] = sum_j P[i][j] (H P)[j][c]
            const double ej = E[j * kNS + c];
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = fma(P[(2 * t + h) * kNS + j], ej, acc[t]);
        }
        wave_sync();
#pragma unroll
        for (int t = 0; t < 16; ++t) K[(2 * t + h) * kNS + c] = acc[t];
        wave_sync();
        // ---- eigen-decomposition: diag(K) = sigma^2, E = V ----------------------------------------------------------
        sym_eig32(K, E, n, c, h);
        wave_sync();
        const double lam = valid ? fmax(K[c * kNS + c], 0.0) : 0.0;
        if (h == 0) Vv[c] = valid ? lam : -1.0;
        wave_sync();
        int rk = 0;                                       // position of my eigenvalue in descending order
        for (int j = 0; j < n; ++j) {
            const double lj = Vv[j];
            rk += (lj > lam || (lj == lam && j < c)) ? 1 : 0;
        }
        const int mm = (m < n) ? m : n;                   // thin SVD: min(rows, columns) singular triplets
        const int live = (mm < rem) ? mm : rem;           // ... of which at most rank(P) are not exactly zero
        const bool in_thin = valid && rk < mm;
        double smax = colmax<32>(valid ? lam : 0.0);
        double sq, rs;
        fast_sqrt_rsqrt(smax > 0.0 ? smax : 1.0, sq, rs);
        smax = smax > 0.0 ? sq : 0.0;
        double sig = 0.0;
        if (lam > 0.0) { fast_sqrt_rsqrt(lam, sq, rs); sig = sq; }
        if (!(rk < live) || sig < kEhqpFloor * smax) sig = 0.0;          // see the header: exact zeros and the unresolved tail
        const double smin = colmin<32>(in_thin ? sig : INFINITY);         // min over the thin set (INFINITY: no rows at all)
        const double thr = fmax(smax * Q.sigma_min, 2.2250738585072014e-308);
        const bool in_rank = in_thin && sig >= thr && sig > 0.0;
        const bool damped = !(smin >= Q.sigma_min);
        const double lam2 = (damped && smin < INFINITY) ? smin * smin : 0.0;
        const double dcoef = in_rank ? fast_rcp(sig * sig + lam2) : 0.0;
        // ---- x += V D V' u, then once more with the residual of that step, u - H dx (refinement of the normal equations:
        // the Gram route alone leaves an error of cond(JP)^2 eps in x, 1e-6 on the worst of a few hundred random stacks)
        double ucur = uvec;
        for (int pass = 0; pass < 2; ++pass) {
            wave_sync();
            if (h == 0) Vv[c] = ucur;
            wave_sync();
            double proj = 0.0;                            // v_c . u over this half's rows
#pragma unroll
            for (int t = 0; t < 16; ++t) proj = fma(E[(2 * t + h) * kNS + c], Vv[2 * t + h], proj);
            proj = halfsum<32>(proj);
            wave_sync();
            if (h == 0) { Vv[c] = dcoef * proj; Vv[32 + c] = in_thin ? 1.0 : 0.0; }
            wave_sync();
            double dx = 0.0;                              // (V t)_c = sum_j E[c][j] t_j
#pragma unroll
            for (int t = 0; t < 16; ++t) dx = fma(E[c * kNS + 2 * t + h], Vv[2 * t + h], dx);
            dx = halfsum<32>(dx);
            if (valid) x += dx;
            if (pass == 0) {                              // u <- u - H dx  (H still in registers: rows 2 t + h, column c)
                wave_sync();
                if (h == 0) Vv[64 + c] = valid ? dx : 0.0;
                wave_sync();
                double hdx = 0.0;
#pragma unroll
                for (int t = 0; t < 16; ++t) hdx = fma(hacc[t], Vv[64 + 2 * t + h], hdx);
                hdx = halfsum<32>(hdx);
                ucur = valid ? ucur - hdx : 0.0;
            }
        }
        // ---- P <- P - V_thin V_thin' ---------------------------------------------------------------------------------
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = 0.0;
        for (int j = 0; j < n; ++j) {
            const double ec = Vv[32 + j] * E[c * kNS + j];
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = fma(E[(2 * t + h) * kNS + j], ec, acc[t]);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) P[(2 * t + h) * kNS + c] -= acc[t];
        wave_sync();
        rem -= live;
        if (Q.x_levels && valid && h == 0) Q.x_levels[(inst * Q.L + k) * n + c] = x;
    }
    if (valid && h == 0) Q.dq[inst * n + c] = x;
    if (lane == 0) {
        Q.status[inst] = QP_SOLVED;                        // eHQP::solve returns true (eHQP.cpp:94)
        if (Q.iterations) Q.iterations[inst] = 0;
    }
}

// validation + kernel arguments from the plan and the per-call batch (shared by the C-ABI entry and tests/emu)
inline int ehqp_args(const osot_plan_desc& p, const osot_qp_batch* b, double sigma_min, bool any_task_inactive, DevEhqp& Q,
                     const char** why) {
    if (p.n > 32) { *why = "eHQP front-end: n <= 32 in this build"; return OSOT_ERR_UNSUPPORTED; }
    if (p.has_regularisation) { *why = "eHQP has no regularisation task"; return OSOT_ERR_UNSUPPORTED; }
    if (any_task_inactive) { *why = "eHQP front-end: Task::setActive(false) is not covered (switch whole levels with level_active)"; return OSOT_ERR_UNSUPPORTED; }
    std::memset(&Q, 0, sizeof(Q));
    Q.B = b->B; Q.n = p.n; Q.L = p.n_levels;
    Q.sigma_min = sigma_min > 0.0 ? sigma_min : 1.0e-12;     // eHQP.cpp:56
    Q.active_mask = 0u;
    for (int k = 0; k < p.n_levels; ++k) {
        int m, ma; plan_level_rows(&p, k, &m, &ma);
        Q.m[k] = m; Q.ma[k] = ma;
        if (ma > 0 && !b->A[k]) { *why = "A[k] is null for a level with stored rows"; return OSOT_ERR_INVALID; }
        if (!b->b[k]) { *why = "b[k] is null"; return OSOT_ERR_INVALID; }
        bool dense = false;
        for (int j = 0; j < p.level[k].n_tasks; ++j) dense = dense || p.level[k].task[j].dense_weight != 0;
        if (dense && (!b->WA[k] || !b->Wb[k])) { *why = "level has a non-diagonal weight but WA[k] / Wb[k] is null"; return OSOT_ERR_INVALID; }
        Q.A[k] = b->A[k]; Q.b[k] = b->b[k]; Q.w[k] = b->w[k];
        Q.WA[k] = dense ? b->WA[k] : nullptr; Q.Wb[k] = dense ? b->Wb[k] : nullptr;
        if (!b->level_active || b->level_active[k]) Q.active_mask |= (1u << k);
    }
    Q.dq = b->dq; Q.status = b->status; Q.iterations = b->iterations; Q.x_levels = b->x_levels;
    return OSOT_OK;
}

}  // namespace osot
