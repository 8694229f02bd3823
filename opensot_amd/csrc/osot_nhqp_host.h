// osot_nhqp_host.h -- host-side sequence of the nHQP front-end (no HIP calls: the three launches come in as functors, so that
// the GPU build and the host emulation of tests/emu run the SAME orchestration).
#pragma once
#include "osot_host_plan.h"
#include "osot_nhqp.h"

namespace osot {

struct NhqpWorkspace {      // per-solver scratch in HBM (host memory under the emulation), sized for max_batch
    double *N[2], *q0, *H, *g, *R, *rlo, *rup, *z, *V2;
    int *qp_status, *qp_iters;
};
// doubles / ints of each array for a batch capacity B
struct NhqpSizes { size_t N, q0, H, g, R, rl, z, V2, st; };
inline NhqpSizes nhqp_sizes(const osot_plan_desc& p, int B) {
    int nc = 0; plan_constraint_rows(&p, &nc);
    const size_t n = p.n, nr = (size_t)nc + n;
    return {B * n * n, B * n, B * n * n, B * n, B * nr * n, B * nr, B * n, B * n * n, (size_t)B};
}

// what the reference's constructor refuses, and what this build does not cover
inline bool nhqp_level_is_wide(int m, int nf) { return (m < nf ? m : nf) > 32; }
// the level's A / b regularisation, selective null-space regularisation and singular-value threshold: the solver-wide setting, or the
// level's own (nHQP::setPerformAbRegularization(level, .), setPerformSelectiveNullSpaceRegularization(level, .),
// setMinSingularValueRatio(vector): nHQP.cpp:127-152, 206-221)
inline void nhqp_level_options(const osot_nhqp_options* opt, int k, int& ab_reg, int& sel_reg, double& thr) {
    ab_reg = !(opt && (opt->no_ab_regularization || opt->level_no_ab_regularization[k]));
    sel_reg = !(opt && (opt->no_selective_ns_regularization || opt->level_no_selective_ns_regularization[k]));
    // nHQP.h:66: 0.05 by default.  A positive value is honoured as it is; the flag is only needed to express 0 ("lift nothing")
    thr = !opt ? 0.05 : (opt->min_sv_ratio_is_set ? opt->min_sv_ratio : (opt->min_sv_ratio > 0.0 ? opt->min_sv_ratio : 0.05));
    if (opt && opt->level_min_sv_ratio_is_set[k]) thr = opt->level_min_sv_ratio[k];
}
inline int nhqp_validate(const osot_plan_desc& p, const osot_nhqp_options* opt, int free_vars[OSOT_MAX_LEVELS], const char** why) {
    if (p.n > OSOT_MAX_VARS) { *why = "nHQP front-end: n <= 64"; return OSOT_ERR_UNSUPPORTED; }
    if (p.has_regularisation) { *why = "nHQP has no regularisation task"; return OSOT_ERR_UNSUPPORTED; }
    for (int j = 0; j < p.n_rowblocks; ++j) {
        if (p.rowblock[j].only_level != 0) { *why = "[nHQP] Local constraints not supported"; return OSOT_ERR_UNSUPPORTED; }   // nHQP.cpp:41-44
        if (rows_are_implicit(p.rowblock[j].kind)) { *why = "nHQP front-end: unit-row blocks are not covered (use the box)"; return OSOT_ERR_UNSUPPORTED; }
    }
    // the singular-value threshold is a ratio: nHQP::setMinSingularValueRatio throws outside [0, 1] (nHQP.cpp:127-152)
    if (opt) {
        auto bad_ratio = [](double v) { return !(v >= 0.0 && v <= 1.0); };   // (NaN fails both comparisons)
        if ((opt->min_sv_ratio_is_set || opt->min_sv_ratio != 0.0) && bad_ratio(opt->min_sv_ratio)) {
            *why = "[nHQP] min_sv_ratio must be in [0, 1]"; return OSOT_ERR_INVALID;
        }
        for (int k = 0; k < p.n_levels; ++k)
            if (opt->level_min_sv_ratio_is_set[k] && bad_ratio(opt->level_min_sv_ratio[k])) {
                *why = "[nHQP] level_min_sv_ratio[k] must be in [0, 1]"; return OSOT_ERR_INVALID;
            }
    }
    int nf = p.n;
    for (int k = 0; k < p.n_levels; ++k) {
        int m, ma; plan_level_rows(&p, k, &m, &ma);
        if (m > 64) { *why = "nHQP front-end: at most 64 rows per level"; return OSOT_ERR_UNSUPPORTED; }
        for (int j = 0; j < p.level[k].n_tasks; ++j)
            if (p.level[k].task[j].dense_weight && !(opt && opt->level_W[k])) {
                *why = "nHQP front-end: a level with a non-diagonal weight needs osot_nhqp_options.level_W[k] (A/b regularisation multiplies the "
                       "REGULARISED A N by W: W A and W b do not suffice)";
                return OSOT_ERR_UNSUPPORTED;
            }
        const int given = opt ? opt->free_vars[k] : 0;
        if (k == 0) { if (given != 0 && given != p.n) { *why = "free_vars[0] must be n"; return OSOT_ERR_INVALID; } }
        else if (given != 0) nf = given;
        if (nf <= 0 || nf > p.n) { *why = "[nHQP] No free variables left at a layer: decrease the number of layers!"; return OSOT_ERR_INVALID; }   // nHQP.cpp:32-35
        // (min(rows, free variables) <= 32: the SVD goes through the 32-wide eigen-decomposition of the small-side Gram matrix; beyond --
        //  round 5 -- through osot_nhqp_prepare_wide_kernel: nhqp_level_is_wide)
        free_vars[k] = nf;
        nf = nf - m;       // default for the next level: full row rank (the constructor's count, nHQP.cpp:88-91, on a full-rank task)
    }
    return OSOT_OK;
}

// prepare(DevNhqp), qp(B, n, nc, H, g, A, lA, uA, l, u, eps, x, status, iters), accumulate(DevNhqpAcc)
// task_active: [OSOT_MAX_LEVELS * OSOT_MAX_TASKS] Task::setActive flags, or null (every task active)
template <class FPrep, class FQp, class FAcc>
int nhqp_run(const osot_plan_desc& p, const osot_qp_batch* b, const osot_nhqp_options* opt, const NhqpWorkspace& ws,
             FPrep prepare, FQp qp, FAcc accumulate, const char** why, const unsigned char* task_active = nullptr) {
    int free_vars[OSOT_MAX_LEVELS];
    int rc = nhqp_validate(p, opt, free_vars, why);
    if (rc != OSOT_OK) return rc;
    int nc = 0; plan_constraint_rows(&p, &nc);
    const int n = p.n, L = p.n_levels, B = b->B;
    const bool has_box = p.n_bounds > 0;
    for (int k = 0; k < L; ++k) {
        int m, ma; plan_level_rows(&p, k, &m, &ma);
        const int nf = free_vars[k];
        const int ns = (k + 1 < L) ? free_vars[k + 1] : ((nf - m > 0) ? nf - m : 0);
        DevNhqp Q;
        std::memset(&Q, 0, sizeof(Q));
        Q.B = B; Q.n = n; Q.nc = nc; Q.level = k; Q.m = m; Q.ma = ma; Q.nf = nf; Q.ns = ns; Q.has_box = has_box ? 1 : 0;
        nhqp_level_options(opt, k, Q.ab_reg, Q.sel_reg, Q.thr);
        if (task_active) {         // rows of the level's inactive tasks (Task::setActive(false): A zeroed, Task.h:383-387)
            int off = 0;
            for (int j = 0; j < p.level[k].n_tasks; ++j) {
                const int rows = p.level[k].task[j].rows;
                if (!task_active[k * OSOT_MAX_TASKS + j])
                    for (int r = off; r < off + rows && r < 64; ++r) Q.zero_rows |= (1ull << r);
                off += rows;
            }
        }
        Q.A = b->A[k]; Q.b = b->b[k]; Q.w = b->w[k];
        Q.Wd = opt ? opt->level_W[k] : nullptr;
        Q.C = b->C; Q.lo = b->lo; Q.up = b->up; Q.l = b->l; Q.u = b->u;
        Q.N = ws.N[k & 1]; Q.q0 = ws.q0;
        Q.H = ws.H; Q.g = ws.g; Q.R = ws.R; Q.rlo = ws.rlo; Q.rup = ws.rup; Q.V2 = ws.V2;
        Q.status = (k == 0) ? nullptr : b->status;
        prepare(Q);
        if (k == 0) rc = qp(B, n, nc, ws.H, ws.g, b->C, b->lo, b->up, has_box ? b->l : nullptr, has_box ? b->u : nullptr, p.eps_abs, ws.z, ws.qp_status, ws.qp_iters);
        else {
            const int nr = nc + (has_box ? n : 0);
            rc = qp(B, nf, nr, ws.H, ws.g, nr ? ws.R : nullptr, nr ? ws.rlo : nullptr, nr ? ws.rup : nullptr, nullptr, nullptr, p.eps_abs, ws.z, ws.qp_status, ws.qp_iters);
        }
        if (rc != OSOT_OK) { *why = "the level's QP launch failed"; return rc; }
        DevNhqpAcc Ac;
        std::memset(&Ac, 0, sizeof(Ac));
        Ac.B = B; Ac.n = n; Ac.nf = nf; Ac.ns = ns; Ac.first = (k == 0); Ac.last = (k == L - 1);
        Ac.z = ws.z; Ac.qp_status = ws.qp_status; Ac.q0 = ws.q0; Ac.N = ws.N[k & 1]; Ac.V2 = ws.V2; Ac.Nnext = ws.N[(k + 1) & 1];
        Ac.status = b->status; Ac.dq = b->dq;
        accumulate(Ac);
    }
    return OSOT_OK;
}

}  // namespace osot
