// osot_qp_core.h -- device-side dense QP core: Cholesky, J = L^-T, dual active set.
//
// Solves, for one instance held by a team of T lanes,
//      min 1/2 x'(H + eps I)x + g'x   s.t.  lo_r <= a_r'x <= up_r  (general rows),  lb <= x <= ub
// which is the problem every level of the cascade hands to OpenSoT's BackEnd
// (include/OpenSoT/solvers/BackEnd.h:125-150; convention SURVEY.md 8b).  H + eps I is strictly convex,
// so the minimiser is unique and any exact method reproduces the reference qpOASES x up to round-off.
//
// Method: Goldfarb-Idnani dual active set (the algorithm of the reference's eiQuadProg back-end,
// external/eiQuadProg-ext/include/eiquadprog.hpp) re-designed for a wavefront:
//   * the n-1 sequential Givens rotations of every constraint addition are replaced by ONE Householder
//     reflection = one rank-1 update of J, fully lane-parallel (w = J2 v comes for free from z);
//   * bilateral rows and the box are native (no [I;-I;A;-A] expansion, eiQuadProgBackEnd.cpp:40-66);
//     box normals are +-e_i, so d = J'n is a row read of J, not a mat-vec;
//   * equalities (lo == up: TaskToConstraint rows and the iHQP optimality rows A_j x = A_j x_j,
//     iHQP.cpp:164-170) are eliminated first and never enter the ratio test; their multipliers are
//     not formed;
//   * the triangular solve r = R^-1 d1 only runs over the inequality part of the working set.
//
// LDS per team (doubles, row stride S = n|1 so that both row- and column-walks are bank-conflict
// free for ds_read_b64):  M1[n][S]  H -> L (Cholesky) -> R (working-set factor),
//                         M2[n][S]  JT, JT[j][k] = J[k][j]  (starts as L^-1),
//                         V[4][T]   broadcast staging vectors.
// Lane t owns element t of x, g, d, z, u and the box state of variable t.
#pragma once
#include <osot_team.h>  // resolved through -I: csrc/ for the product, tests/emu/ for the host emulation

namespace osot {

constexpr double kInfty = 1.0e20;      // QPOasesBackEnd::checkINFTY clamp (QPOasesBackEnd.cpp:339-356)
constexpr double kDepTol2 = 1.0e-18;   // |d2|^2 <= kDepTol2 |d|^2  -> normal is in the span of the working set
constexpr double kViolTol = 1.0e-11;   // a slack below -kViolTol*max(1,|bound|) counts as violated
constexpr double kEqTol = 1.0e-9;      // consistency of a linearly dependent equality row

enum { QP_SOLVED = 0, QP_INFEASIBLE = 1, QP_MAX_ITER = 2, QP_NOT_PD = 3 };

template <int T>
struct TeamCtx {
    int tl;         // lane within the team
    int n, S;
    double* M1;     // n*S
    double* M2;     // n*S
    double* V;      // 4*T
    int* rowstate;  // one int per general row: 0 free, 1 lower active, 2 upper active, 3 equality
};

__device__ __forceinline__ double clamp_inf(double v) {
    return v < -kInfty ? -kInfty : (v > kInfty ? kInfty : v);
}

// One Householder reflection that maps d2 = d[iq:] onto alpha*e_iq, applied to the columns iq.. of J
// (rows iq.. of JT); appends (d1; alpha) as column iq of R.  z = J2 d2 is an input.
template <int T>
__device__ __forceinline__ void householder_add(const TeamCtx<T>& c, double d, double d2, double z,
                                                double nd2, int iq) {
    const int tl = c.tl, n = c.n, S = c.S;
    double* V2 = c.V + 2 * T;
    const double d_iq = team_bcast<T>(d, iq);
    const double nrm = sqrt(nd2);
    const double alpha = (d_iq > 0.0) ? -nrm : nrm;
    const double v = (tl > iq) ? d2 : ((tl == iq) ? d_iq - alpha : 0.0);
    const double beta = 1.0 / (nd2 - alpha * d_iq);
    V2[tl] = v * beta;
    team_sync();
    if (tl < n) {
        const double w = z - alpha * c.M2[iq * S + tl];   // w = J2 v
        for (int j = iq; j < n; ++j) c.M2[j * S + tl] -= V2[j] * w;
        if (tl < iq) c.M1[tl * S + iq] = d;
        else if (tl == iq) c.M1[iq * S + iq] = alpha;
    }
    team_sync();
}

// Remove working-set position qq: shift R, re-triangularise with Givens rotations applied to the rows
// of R and to the matching columns of J (eiquadprog.hpp:551-617 does the same on its storage).
template <int T>
__device__ __forceinline__ void drop_constraint(const TeamCtx<T>& c, int qq, int& iq, int& Aq, double& uq) {
    const int tl = c.tl, n = c.n, S = c.S;
    const int An = team_shift_down_i<T>(Aq);
    const double un = team_shift_down<T>(uq);
    if (tl >= qq && tl < iq - 1) { Aq = An; uq = un; }
    if (tl < n) {
        for (int q = qq; q < iq - 1; ++q)
            if (tl <= q + 1) c.M1[tl * S + q] = c.M1[tl * S + q + 1];
    }
    team_sync();
    iq--;
    for (int j = qq; j < iq; ++j) {
        const double a = c.M1[j * S + j], b = c.M1[(j + 1) * S + j];
        team_sync();   // every lane has read the pivot pair before lane j overwrites it
        const double h = sqrt(a * a + b * b);
        if (h == 0.0) continue;
        const double cg = a / h, sg = b / h;
        if (tl >= j && tl < iq) {
            const double r1 = c.M1[j * S + tl], r2 = c.M1[(j + 1) * S + tl];
            c.M1[j * S + tl] = cg * r1 + sg * r2;
            c.M1[(j + 1) * S + tl] = cg * r2 - sg * r1;
        }
        if (tl < n) {
            const double j1 = c.M2[j * S + tl], j2 = c.M2[(j + 1) * S + tl];
            c.M2[j * S + tl] = cg * j1 + sg * j2;
            c.M2[(j + 1) * S + tl] = cg * j2 - sg * j1;
        }
        team_sync();
    }
}

// RowSrc: double elem(int r, int lane) (a_r[lane], 0 for lane >= n), double lo(int r), double up(int r)
// Pre: M1 holds H + eps I (lower triangle used).  Post: returns status, x is lane-distributed.
template <int T, class RowSrc>
__device__ int gi_solve(const TeamCtx<T>& c, const RowSrc& rows, int nrows, double g, bool has_box,
                        double lb, double ub, int max_iter, double& x_out, int& iters_out) {
    const int tl = c.tl, n = c.n, S = c.S;
    double* M1 = c.M1;
    double* M2 = c.M2;
    double* V0 = c.V;
    double* V1 = c.V + T;
    const bool valid = tl < n;
    lb = clamp_inf(lb);
    ub = clamp_inf(ub);

    // ---- Cholesky H + eps I = L L' in place (lane = row) -------------------------------------
    for (int j = 0; j < n; ++j) {
        double s = 0.0;
        if (valid && tl >= j) {
            s = M1[tl * S + j];
            for (int k = 0; k < j; ++k) s -= M1[tl * S + k] * M1[j * S + k];
        }
        const double piv = team_bcast<T>(s, j);
        if (!(piv > 0.0)) { x_out = 0.0; iters_out = 0; return QP_NOT_PD; }
        const double ljj = sqrt(piv);
        if (valid && tl >= j) M1[tl * S + j] = (tl == j) ? ljj : s / ljj;
        team_sync();
    }
    // ---- JT = L^-1 into M2 (lane = column) ----------------------------------------------------
    double invd = 0.0;   // lane i: 1 / L[i][i]
    if (valid) {
        for (int i = 0; i < n; ++i) {
            double y = 0.0;
            if (i == tl) { y = 1.0 / M1[i * S + i]; invd = y; }
            else if (i > tl) {
                double s = 0.0;
                for (int k = tl; k < i; ++k) s += M1[i * S + k] * M2[k * S + tl];
                y = -s / M1[i * S + i];
            }
            M2[i * S + tl] = y;
        }
    }
    team_sync();
    // ---- unconstrained minimiser: L y = -g, L' x = y by substitution ----------------------------
    // (NOT x = -J J'g: with H = A'A + eps I rank deficient, g lies in range(A') and the substitution
    //  keeps the exact cancellation in the eps-pivots that the explicit inverse loses -- the
    //  reference's own known-answer test TestQPOases.cpp:274-340 needs it at 1e-6)
    double x = valid ? -g : 0.0;
    for (int j = 0; j < n; ++j) {
        const double yj = team_bcast<T>(x * invd, j);
        if (tl == j) x = yj;
        else if (valid && tl > j) x -= M1[tl * S + j] * yj;
    }
    for (int j = n - 1; j >= 0; --j) {
        const double xj = team_bcast<T>(x * invd, j);
        if (tl == j) x = xj;
        else if (tl < j) x -= M1[j * S + tl] * xj;
    }
    team_sync();

    int iq = 0;          // size of the working set
    int Aq = -1;         // lane q < iq: code of the constraint at working-set position q
    double uq = 0.0;     // lane q < iq: its multiplier (inequalities only)
    int box_state = 0;   // lane i: 0 free, 1 lower bound active, 2 upper bound active
    int iters = 0;

    // ---- equalities first ----------------------------------------------------------------------
    for (int r = 0; r < nrows; ++r) {
        const double lo = clamp_inf(rows.lo(r)), up = clamp_inf(rows.up(r));
        const bool is_eq = (lo == up) && (lo > -kInfty) && (lo < kInfty);
        if (tl == 0) c.rowstate[r] = is_eq ? 3 : 0;
        if (!is_eq) continue;
        const double a = rows.elem(r, tl);
        V0[tl] = a;
        team_sync();
        double d = 0.0;
        if (valid) for (int k = 0; k < n; ++k) d += M2[tl * S + k] * V0[k];
        const double d2 = (tl >= iq) ? d : 0.0;
        const double dd = team_sum<T>(d * d);
        const double nd2 = team_sum<T>(d2 * d2);
        const double resid = lo - team_sum<T>(a * x);
        if (!(nd2 > kDepTol2 * dd)) {   // row is (numerically) a combination of the rows already in
            if (fabs(resid) <= kEqTol * fmax(1.0, fabs(lo))) continue;   // redundant and consistent
            x_out = x; iters_out = iters; return QP_INFEASIBLE;
        }
        V1[tl] = d2;
        team_sync();
        double z = 0.0;
        if (valid) for (int j = iq; j < n; ++j) z += M2[j * S + tl] * V1[j];
        x += (resid / nd2) * z;
        householder_add<T>(c, d, d2, z, nd2, iq);
        if (tl == iq) Aq = -2 - r;
        iq++;
        iters++;
    }
    const int me = iq;
    team_sync();

    // ---- inequality loop -------------------------------------------------------------------------
    int status = QP_SOLVED;
    const int kNone = 0x7fffffff;
    for (;;) {
        // most violated constraint outside the working set (eiquadprog.hpp:300-315 picks the same)
        double cand = 0.0;
        int code = kNone;
        if (has_box && valid) {
            if (box_state != 1 && lb > -kInfty) {
                const double s = x - lb;
                if (s < -kViolTol * fmax(1.0, fabs(lb)) && s < cand) { cand = s; code = tl; }
            }
            if (box_state != 2 && ub < kInfty) {
                const double s = ub - x;
                if (s < -kViolTol * fmax(1.0, fabs(ub)) && s < cand) { cand = s; code = n + tl; }
            }
        }
        for (int r = 0; r < nrows; ++r) {
            const int st = c.rowstate[r];
            if (st == 3) continue;
            const double lo = clamp_inf(rows.lo(r)), up = clamp_inf(rows.up(r));
            const bool has_lo = lo > -kInfty, has_up = up < kInfty;
            if (!has_lo && !has_up) continue;
            const double ax = team_sum<T>(rows.elem(r, tl) * x);
            if (st != 1 && has_lo) {
                const double s = ax - lo;
                if (s < -kViolTol * fmax(1.0, fabs(lo)) && s < cand) { cand = s; code = 2 * n + 2 * r; }
            }
            if (st != 2 && has_up) {
                const double s = up - ax;
                if (s < -kViolTol * fmax(1.0, fabs(up)) && s < cand) { cand = s; code = 2 * n + 2 * r + 1; }
            }
        }
        team_argmin<T>(cand, code);
        if (code == kNone) break;   // primal feasible: optimal
        if (++iters > max_iter) { status = QP_MAX_ITER; break; }

        const int ip = code;
        double s_ip = cand;
        double u_new = 0.0;
        const bool ip_box = ip < 2 * n;
        const int ip_var = ip_box ? (ip < n ? ip : ip - n) : 0;
        const int ip_row = ip_box ? 0 : (ip - 2 * n) >> 1;
        const double ip_sgn = ip_box ? (ip < n ? 1.0 : -1.0) : ((ip & 1) ? -1.0 : 1.0);
        double np = 0.0;   // lane-distributed normal (general rows only)
        if (!ip_box) np = ip_sgn * rows.elem(ip_row, tl);

        bool failed = false;
        for (;;) {
            // d = J' n
            double d = 0.0;
            if (ip_box) {
                if (valid) d = ip_sgn * M2[tl * S + ip_var];
            } else {
                V0[tl] = np;
                team_sync();
                if (valid) for (int k = 0; k < n; ++k) d += M2[tl * S + k] * V0[k];
            }
            const double d2 = (tl >= iq) ? d : 0.0;
            const double dd = team_sum<T>(d * d);
            const double nd2 = team_sum<T>(d2 * d2);
            const bool z_ok = nd2 > kDepTol2 * dd;
            // z = J2 d2 : primal step direction
            V1[tl] = d2;
            team_sync();
            double z = 0.0;
            if (valid) for (int j = iq; j < n; ++j) z += M2[j * S + tl] * V1[j];
            // r = R^-1 d1 restricted to the inequality part [me, iq): dual step direction
            double rr = 0.0;
            {
                double d1 = (tl < iq) ? d : 0.0;
                for (int j = iq - 1; j >= me; --j) {
                    const double rj = team_bcast<T>(d1, j) / M1[j * S + j];
                    if (tl == j) rr = rj;
                    if (tl >= me && tl < j) d1 -= M1[tl * S + j] * rj;
                }
            }
            // step lengths (eiquadprog.hpp:343-366)
            double t1 = (tl >= me && tl < iq && rr > 0.0) ? fmax(uq, 0.0) / rr : INFINITY;
            int lpos = tl;
            team_argmin<T>(t1, lpos);
            const double t2 = z_ok ? (-s_ip / nd2) : INFINITY;
            if (!(t1 < INFINITY) && !(t2 < INFINITY)) { failed = true; break; }   // infeasible
            if (t2 <= t1) {
                // full step: constraint ip becomes active
                x += t2 * z;
                if (tl >= me && tl < iq) uq -= t2 * rr;
                u_new += t2;
                householder_add<T>(c, d, d2, z, nd2, iq);
                if (tl == iq) { Aq = ip; uq = u_new; }
                if (ip_box) { if (tl == ip_var) box_state = (ip < n) ? 1 : 2; }
                else { if (tl == 0) c.rowstate[ip_row] = (ip & 1) ? 2 : 1; }
                iq++;
                team_sync();
                break;
            }
            // partial step (or pure dual step when z == 0): drop the blocking constraint
            if (z_ok) x += t1 * z;
            if (tl >= me && tl < iq) uq -= t1 * rr;
            u_new += t1;
            const int cdrop = team_bcast_i<T>(Aq, lpos);
            if (cdrop < 2 * n) { if (tl == (cdrop < n ? cdrop : cdrop - n)) box_state = 0; }
            else { if (tl == 0) c.rowstate[(cdrop - 2 * n) >> 1] = 0; }
            drop_constraint<T>(c, lpos, iq, Aq, uq);
            if (z_ok) {
                if (ip_box) s_ip = team_bcast<T>((ip < n) ? (x - lb) : (ub - x), ip_var);
                else {
                    const double ax = team_sum<T>(np * x);   // = sgn * a'x
                    s_ip = (ip & 1) ? (clamp_inf(rows.up(ip_row)) + ax) : (ax - clamp_inf(rows.lo(ip_row)));
                }
            }
            if (++iters > max_iter) { status = QP_MAX_ITER; failed = true; break; }
        }
        if (failed) { if (status == QP_SOLVED) status = QP_INFEASIBLE; break; }
    }
    x_out = x;
    iters_out = iters;
    return status;
}

}  // namespace osot
